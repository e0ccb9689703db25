"""bench.py's multi-GPU launch logic, on the CPU: `--gpus N` must never degrade silently to one rank
(`--gpus 8` without WORLD_SIZE ran ONE rank and printed n_gpus: 1)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra)
    return env


def test_launch_command_is_one_rank_per_gpu_on_localhost():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(["--gpus", "4", "--steps", "3"], 4, port=29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5] == BENCH and cmd[-4:] == ["--gpus", "4", "--steps", "3"]


def test_more_gpus_than_the_node_has_fails_loudly():
    # (this container has no GPU; on a 1-GPU box the same path refuses --gpus 2)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 2
    assert "refusing to run fewer ranks" in r.stderr
    assert "n_gpus" not in r.stdout


def test_world_size_must_equal_gpus():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "n_gpus" not in r.stdout
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and "n_gpus" not in r.stdout


def test_eight_ranks_compose_and_refuse_on_fewer_devices():
    """`--gpus 8` (the driver's scaling run): the launch command is 8 ranks on 127.0.0.1, and a node with fewer than 8
    visible GPUs refuses (exit code 2, no JSON line) instead of printing a smaller job's number."""
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, port=29533)
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29533"
    assert cmd[-7] == BENCH and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 2
    assert "refusing to run fewer ranks" in r.stderr and "n_gpus" not in r.stdout
