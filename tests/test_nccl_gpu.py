"""The RCCL branch of the path's one exchange step on real hardware (missing #4: "the nccl branch of
AllGatherImageFeatures.backward and init_process_group("nccl", device_id=...) have never executed on any machine").
The GPU box has ONE GPU: a process group of world size 1 still goes through RCCL's communicator set-up,
``all_gather_into_tensor`` and ``reduce_scatter_tensor`` -- the calls the N-GPU bench makes; the multi-rank semantics
(block layout, an image-less rank, padding) are held by the world-size-2 gloo tests (tests/test_distributed.py)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "mm-interleaved_amd")]
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)             # (bench.py's call)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
from mmfs_amd import bank
g = torch.Generator(device=dev).manual_seed(1)
local = torch.randn(5, 84, 64, device=dev, generator=g).to(torch.bfloat16).requires_grad_(True)
out = bank.AllGatherImageFeatures.apply(local, None)       # all_gather_into_tensor
assert out.shape == local.shape and torch.equal(out, local.detach())
go = torch.randn(out.shape, device=dev, generator=g).to(torch.bfloat16)
out.backward(go)                                           # reduce_scatter_tensor (the nccl branch)
assert torch.equal(local.grad, go)
loss = bank.keep_in_graph(torch.zeros((), device=dev), out.detach().requires_grad_(True))
t = torch.ones(3, device=dev)
dist.all_reduce(t)
assert float(t.sum()) == 3.0
dist.barrier()
dist.destroy_process_group()
print("nccl ok")
'''


@pytest.mark.gpu
def test_exchange_step_on_rccl_world_size_1():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "nccl ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
