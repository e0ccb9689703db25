#!/usr/bin/env python3
"""bench.py -- MMFS ms_deform_attn forward+backward throughput on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N = 1: plain process.  N > 1: launched by torch.distributed.run, one rank per GPU
  (RCCL) -- by the driver, or by bench.py itself when it finds no WORLD_SIZE (it re-executes under
  torch.distributed.run on 127.0.0.1; fewer visible GPUs than N is an error, never a silent 1-rank run).
  W untimed warm-up steps, then exactly K timed steps bracketed by
  barrier + torch.cuda.synchronize() on both sides, MAX over ranks, rank 0 prints ONE
  JSON line.

Workload (BASELINE.json configs[1] / north_star synthetic tensors, SURVEY.md 8d
"Config 2"):  per GPU  B=8, Nq=4096, n_levels=4 (64^2,32^2,16^2,8^2 -> S=5440), n_heads=8,
n_points=4, C=1024 (D=128), bf16 storage / fp32 arithmetic, synthetic inputs in the
reference's test distribution, already resident in HBM.  One step = one forward +
one backward of the op through the drop-in boundary (MSDeformAttnFunction ->
MultiScaleDeformableAttention shim -> C ABI -> gfx950 kernels), including the
zero-fill and the fp32->bf16 cast of grad_value that belong to the op.

Multi-GPU: the path shards on the batch axis with no data-path collective
(SURVEY.md 8e); every rank processes its own B samples -> "scaling": "weak";
value = N * B * K / t_max.

The K timed steps run clean (one C call per pass, no events).  Per-kernel times come from a
SEPARATE pass of --event-steps steps before the warm-up (HIP events recorded on the launch stream
around every C-ABI launch), so ms_per_step does not depend on the sampling.  Order of a run: inputs,
parity guard, event pass, W warm-up steps, barrier + synchronise, K timed steps, barrier + synchronise.

Extra objects on the JSON line:
  roofline     dominant kernel's algorithmic bytes per launch / its mean duration (the event
               pass) vs 8 TB/s; "traffic" = HBM bytes of that kernel from profiles/pmc_traffic.json
               when that file has an entry for this workload and kernel ("traffic_from" names the
               rocprofv3 --pmc run it was taken from), else null
  kernels_frac the same fraction for EVERY kernel of the step (its own algorithmic bytes / its mean duration / 8 TB/s):
               with three kernels within a few us of each other the "dominant" one changes from box to box
  exchange     N > 1 only: the path's one exchange step, timed after the main region:
               every rank all-gathers the multiscale features of its block of context images (LLM
               geometry, 4 images per sequence: BASELINE config 5) and builds its sequences' bank
               from the gathered tensor -- us per all-gather, GB/s per xGMI link, us per bank build
  step_roofline  the WHOLE step: SURVEY 8d's fwdbwd_bytes / the clean step time vs 8 TB/s ("frac") and vs the 6.29 TB/s a
               plain copy reaches on this part ("frac_of_copy_ceiling"); forward and backward separately from the event pass
  fp16         (default line only, N = 1) the same shape in fp16 -- the only 16-bit type the reference's op has: ms per step
               and the largest error of a slab of ITS results against the CPU oracle (bar: north_star's 1e-3)
  dropin_unchanged_ms  (default line only, N = 1) ms per step when spatial_shapes / level_start_index are rebuilt per call as
               the reference's callers do (what an unchanged reference gets; "ms_per_step" uses tables built once)
  cpu_baseline the oracle's restatement of the reference's only CPU path
               (ms_deform_attn_core_pytorch) timed on this host at BASELINE config 1
               (B=2, Nq=1024, L=4, H=8, P=4, C=256, fp32), all cores and 1 thread, rank 0, N=1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "mm-interleaved_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
HBM_COPY_GBS = 6290.0          # what a float4 copy measures on this part (same guide: 79 % of the data sheet)

WORKLOADS = {
    # name: (B, Nq, H, D, P, per-image level shapes, n_images, dtype)
    "cfg2_northstar": dict(B=8, Nq=4096, H=8, D=128, P=4,
                           shapes=[(64, 64), (32, 32), (16, 16), (8, 8)], n=1, dtype="bf16"),
    "cfg1": dict(B=2, Nq=1024, H=8, D=32, P=4,
                 shapes=[(64, 64), (32, 32), (16, 16), (8, 8)], n=1, dtype="f32"),
    "cfg2_sd_real": dict(B=8, Nq=4096, H=16, D=64, P=8,
                         shapes=[(64, 64), (32, 32), (16, 16), (8, 8)], n=1, dtype="bf16"),
    "cfg5_llm_n4": dict(B=4, Nq=2048, H=16, D=64, P=8,
                        shapes=[(32, 32), (16, 16), (8, 8)], n=4, dtype="bf16"),
    # BASELINE config 3's op: one image per sequence, the LLM layer's geometry (modeling_llama_mmfs.py:326-339), 2048 tokens
    "cfg3_llm_n1": dict(B=4, Nq=2048, H=16, D=64, P=8,
                        shapes=[(32, 32), (16, 16), (8, 8)], n=1, dtype="bf16"),
    # the ViT-Adapter's encoder-side calls at 224 px (vit_adapter_hf.py:112-133,
    # adapter_modules.py:30-49: ViT-L, deform_ratio 0.5 -> D=32, P=4), 32 images per GPU: the injector
    # (256 ViT tokens sample the 32^2/16^2/8^2 pyramid) and the extractor (the 1344 pyramid tokens
    # sample the 16^2 ViT map); 5 of each per ViT forward
    "enc_injector": dict(B=32, Nq=256, H=16, D=32, P=4, shapes=[(32, 32), (16, 16), (8, 8)], n=1, dtype="bf16"),
    "enc_extractor": dict(B=32, Nq=1344, H=16, D=32, P=4, shapes=[(16, 16)], n=1, dtype="bf16"),
    # the only workload the reference itself ever timed: ops/tests/speed_test.py:67-88 (bs 32, two levels
    # 16^2 / 8^2, 128 queries, 64 points per level, 8 heads of 128 channels; fp16, then fp32 via --dtype f32;
    # its loss is .sum(): --grad ones)
    "ref_speed_test": dict(B=32, Nq=128, H=8, D=128, P=64, shapes=[(16, 16), (8, 8)], n=1, dtype="f16"),
}
DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


def algorithmic_bytes(w, e):
    """compulsory traffic, every distinct tensor element touched once."""
    B, Nq, H, D, P = w["B"], w["Nq"], w["H"], w["D"], w["P"]
    Leff = len(w["shapes"]) * w["n"]
    S = sum(h * ww for h, ww in w["shapes"]) * w["n"]
    C = H * D
    pts = B * Nq * H * Leff * P
    fwd = e * (B * S * C + 3 * pts + B * Nq * C)
    bwd = e * (2 * B * S * C + 6 * pts + B * Nq * C)
    # the staged backward: each kernel priced on the tensors IT must touch once (scratch excluded)
    taps = e * (B * S * C + 6 * pts + B * Nq * C)       # value, loc, attn, grad_out -> grad_loc, grad_attn
    # hybrid routing (csrc/msda_dense.hip): levels of <= min(256, 64 P) pixels get their grad_loc /
    # grad_attn from dense dot products on the matrix cores, the others from the row-gather kernel
    dense = [h * ww <= min(256, 64 * P) for h, ww in w["shapes"]] * w["n"]
    Sd = sum(h * ww for (h, ww), dn in zip(w["shapes"] * w["n"], dense) if dn)
    pts_d = B * Nq * H * sum(dense) * P
    taps_dense = e * (B * Sd * C + 6 * pts_d + B * Nq * C)
    taps_fine = e * (B * (S - Sd) * C + 6 * (pts - pts_d) + B * Nq * C)
    # grad_value: the cell sort (loc, attn -> scratch) + tile reduce
    sort = e * 3 * pts
    red = e * (B * S * C + B * Nq * C)                  # grad_out -> grad_value
    return dict(msda_fwd=fwd, msda_bwd_atomic=bwd, msda_bwd_taps=taps, msda_bwd_value_sort=sort,
                msda_bwd_value_reduce=red, fwdbwd=fwd + bwd, bwd=bwd,
                msda_bwd_taps_coarse=taps_dense, msda_bwd_taps_fine=taps_fine)


def make_inputs(w, device, seed, loc_dist="uniform", visible="all"):
    g = torch.Generator(device=device).manual_seed(seed)
    dt = DTYPES[w["dtype"]]
    shapes = torch.tensor(w["shapes"] * w["n"], dtype=torch.long, device=device)
    start = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    B, Nq, H, D, P = w["B"], w["Nq"], w["H"], w["D"], w["P"]
    S, L = int(shapes.prod(1).sum()), shapes.shape[0]
    value = torch.rand(B, S, H, D, device=device, generator=g).to(dt)
    loc = torch.rand(B, Nq, H, L, P, 2, device=device, generator=g)
    if loc_dist == "centre":
        # the LLM path's distribution: every query samples within a few pixels of ONE reference point
        px = (shapes.float().flip(-1)).view(1, 1, 1, L, 1, 2)               # (W, H) per level
        loc = 0.5 + (loc - 0.5) * 16.0 / px                                    # +-8 pixels around the centre
    loc = loc.to(dt)
    attn = torch.rand(B, Nq, H, L, P, device=device, generator=g) + 1e-5
    if visible == "causal" and w["n"] > 1:
        # the LLM path's mask (mm_interleaved.py:199-221): token q sees image k only if the image
        # comes before it -- here image k of n enters at query k * Nq / n; what a token cannot see
        # gets exactly 0 from MMFS's masked softmax (mmfs.py:203-231)
        img = torch.arange(L, device=device) // (L // w["n"])
        vis = torch.arange(Nq, device=device)[:, None] * w["n"] >= img[None, :] * Nq            # [Nq, L]
        attn = attn * vis[None, :, None, :, None]
    attn = (attn / attn.sum((-1, -2), keepdim=True)).to(dt)
    grad = torch.randn(B, Nq, H * D, device=device, generator=g).to(dt)
    return value, shapes, start, loc, attn, grad


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(budget_s=12.0):
    """Times oracle/msda_torch.py (restatement of the reference's CPU path, checked against the imported
    reference's goldens in tests/) at BASELINE config 1, fp32, forward + backward, the discipline of the
    reference's ops/tests/speed_test.py:30-55 scaled to a CPU (5 warm-up iterations, then timed
    iterations until ``budget_s`` or 20 of them): once with every host core, once with one thread."""
    from oracle import msda_torch
    w = WORKLOADS["cfg1"]
    g = torch.Generator().manual_seed(0)
    shapes = w["shapes"] * w["n"]
    B, Nq, H, D, P = w["B"], w["Nq"], w["H"], w["D"], w["P"]
    S, L = sum(h * ww for h, ww in shapes), len(shapes)
    value = torch.rand(B, S, H, D, generator=g)
    loc = torch.rand(B, Nq, H, L, P, 2, generator=g)
    attn = torch.rand(B, Nq, H, L, P, generator=g) + 1e-5
    attn = attn / attn.sum((-1, -2), keepdim=True)
    grad = torch.ones(B, Nq, H * D)
    all_threads = torch.get_num_threads()
    runs = {}
    for threads in (all_threads, 1):
        torch.set_num_threads(threads)
        for _ in range(5 if threads > 1 else 1):
            msda_torch.fwd_bwd(value, shapes, loc, attn, grad)
        iters, t0 = 0, time.perf_counter()
        while True:
            msda_torch.fwd_bwd(value, shapes, loc, attn, grad)
            iters += 1
            el = time.perf_counter() - t0
            if el > budget_s / 2 or iters >= 20:
                break
        runs[threads] = (B * iters / el, el / iters * 1e3, iters)
    torch.set_num_threads(all_threads)
    ab = algorithmic_bytes(dict(w), 4)["fwdbwd"]
    # "value" is the better of the two runs (on a 128-core host the all-cores run of this small problem
    # is often the slower one); both are on the line
    best_threads = all_threads if runs[all_threads][0] >= runs[1][0] else 1
    best = runs[best_threads]
    return {"value": round(best[0], 3), "unit": "samples/s", "cores": best_threads, "kind": "port",
            "ms_per_iter": round(best[1], 2), "effective_GBs": round(ab / (best[1] * 1e-3) / 1e9, 3),
            "all_cores": {"threads": all_threads, "value": round(runs[all_threads][0], 3), "ms_per_iter": round(runs[all_threads][1], 2)},
            "one_thread": {"value": round(runs[1][0], 3), "ms_per_iter": round(runs[1][1], 2)},
            "cpu": cpu_model(), "host_cpus": os.cpu_count(),
            "sample": f"oracle/msda_torch.py (ms_deform_attn_core_pytorch restated), BASELINE config 1 "
                      f"(B={B} Nq={Nq} L={L} H={H} P={P} C={H * D}, fp32, fwd+bwd), {best[2]} timed iterations with "
                      f"{all_threads} threads and {runs[1][2]} with 1 thread after warm-up"}


def timed_steps(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def fp16_sibling(w, device, shapes, start, steps=20, warmup=10):
    """The headline shape in fp16 -- the only 16-bit type the reference's op has (ms_deform_attn_cuda.cu:65 dispatches
    fp64 / fp32 / fp16; training is fp16, configs/release/mm_pretrain.yaml:3) and so the only one in which BASELINE's
    "within 1e-3 of the reference" can be checked at the headline shape: ms per step, and the largest error of a slab of
    the results against the CPU oracle (the checker of tests/, here on the very tensors that were timed): a strided set of
    48 queries of the first, a middle and the last sample (out, grad_loc, grad_attn: every head, every level) and
    grad_value of one whole head of sample 0.  Errors are absolute for ``out`` (values in [0, 1)) and relative to the
    largest reference entry for the gradients."""
    from mmfs_amd.functions import MSDeformAttnFunction
    from oracle import msda_oracle
    w16 = dict(w, dtype="f16")
    value, _, _, loc, attn, grad = make_inputs(w16, device, seed=0)
    value.requires_grad_(True); loc.requires_grad_(True); attn.requires_grad_(True)

    def step():
        out = MSDeformAttnFunction.apply(value, shapes, start, loc, attn, 1)
        return out, torch.autograd.grad(out, (value, loc, attn), grad)

    ms = timed_steps(step, steps, warmup)
    out, (gv, gl, ga) = step()
    torch.cuda.synchronize()
    B, Nq, H, D = w["B"], w["Nq"], w["H"], w["D"]
    sh, st = shapes.cpu(), start.cpu()
    err = {"out": 0.0, "grad_loc": 0.0, "grad_attn": 0.0, "grad_value": 0.0}
    rel = lambda got, want: float(np.abs(got.double().cpu().numpy() - np.asarray(want, np.float64).reshape(got.shape)).max()
                                  / max(1.0, float(np.abs(np.asarray(want)).max())))
    for bi in sorted({0, B // 2, B - 1}):
        qs = torch.arange(5, Nq, max(1, Nq // 48), device=device)[:48]
        x = [t.detach().double().cpu() for t in (value[bi:bi + 1], loc[bi:bi + 1, qs], attn[bi:bi + 1, qs], grad[bi:bi + 1, qs])]
        want = msda_oracle.forward(x[0], sh, st, x[1], x[2])
        _, wgl, wga = msda_oracle.backward(x[0], sh, st, x[1], x[2], x[3])
        err["out"] = max(err["out"], float(np.abs(out[bi:bi + 1, qs].detach().double().cpu().numpy() - np.asarray(want).reshape(1, len(qs), -1)).max()))
        err["grad_loc"] = max(err["grad_loc"], rel(gl[bi:bi + 1, qs], wgl))
        err["grad_attn"] = max(err["grad_attn"], rel(ga[bi:bi + 1, qs], wga))
    h0 = 3 % H
    x = [t.detach().double().cpu() for t in (value[:1, :, h0:h0 + 1], loc[:1, :, h0:h0 + 1], attn[:1, :, h0:h0 + 1],
                                               grad[:1, :, h0 * D:(h0 + 1) * D])]
    wgv, _, _ = msda_oracle.backward(x[0], sh, st, x[1], x[2], x[3])
    err["grad_value"] = rel(gv[:1, :, h0:h0 + 1], wgv)
    return {"dtype": "f16", "ms_per_step": round(ms, 4), "steps": steps,
            "samples_per_s": round(w["B"] / (ms * 1e-3), 2),
            "max_err_vs_oracle_slab": {k: float(f"{v:.3e}") for k, v in err.items()},
            "bar": "1e-3 (north_star, fp16): absolute for out, relative to the largest reference entry for the gradients",
            "within_bar": bool(max(err.values()) <= 1e-3),
            "slab": "48 strided queries of samples 0, B/2, B-1 (out, grad_loc, grad_attn: all heads and levels); grad_value of head 3 of sample 0"}


class ExchangeOverlap:
    """the feature all-gather overlapped with the op.  issue() launches the RCCL all-gather of this
    rank's image block on a side stream (after the caller's stream has reached this point), join() makes the
    caller's stream wait for it -- the bank build that consumes it belongs to the NEXT module call."""

    def __init__(self, device, rank, world):
        from mmfs_amd import bank
        self.bank = bank
        B_local, n, hw, C = 4, 4, 1344, 1024
        self.n_img = world * B_local * n
        per_rank = bank.images_per_rank(self.n_img, world)
        g = torch.Generator(device=device).manual_seed(100 + rank)
        self.mine = torch.randn(per_rank, hw, C, device=device, generator=g).to(torch.bfloat16)
        self.side = torch.cuda.Stream(device=device)
        self.device = device
        self.out = None

    def issue(self):
        main = torch.cuda.current_stream(self.device)
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            self.out = self.bank.all_gather_image_features(self.mine, self.n_img)

    def join(self):
        torch.cuda.current_stream(self.device).wait_stream(self.side)


def exchange_step(device, rank, world, dist, steps=20):
    """The path's one exchange step at BASELINE config 5's geometry: 4 context images per sequence, LLM
    pyramid (32^2, 16^2, 8^2 -> 1344 tokens) x C = 1024 bf16 = 2.75 MB per image; every rank encodes a
    contiguous block of the images of ALL sequences and needs the images of ITS sequences.  Timed: the
    RCCL all-gather of the packed features, and the bank build (one index-gather) from its result."""
    from mmfs_amd import bank
    B_local, n, hw, C = 4, 4, 1344, 1024
    n_img = world * B_local * n
    per_rank = bank.images_per_rank(n_img, world)
    g = torch.Generator(device=device).manual_seed(100 + rank)
    mine = torch.randn(per_rank, hw, C, device=device, generator=g).to(torch.bfloat16)
    num = torch.full((B_local,), n, device=device)
    first = rank * B_local * n
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_gather = t_bank = 0.0
    for i in range(steps + 3):
        ev[0].record()
        allf = bank.all_gather_image_features(mine, n_img)
        ev[1].record()
        bk = bank.llm_feature_bank(allf[first:first + B_local * n], num, n)
        ev[2].record()
        torch.cuda.synchronize()
        if i >= 3:
            t_gather += ev[0].elapsed_time(ev[1]); t_bank += ev[1].elapsed_time(ev[2])
    assert bk.shape == (B_local, n, hw, C)
    t = torch.tensor([t_gather / steps, t_bank / steps], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    shard = per_rank * hw * C * 2
    us = float(t[0]) * 1e3
    return {"what": f"all_gather_into_tensor of {per_rank} images x {hw} tokens x {C} ch bf16 per rank + bank build "
                    f"for {B_local} sequences x {n} images (BASELINE config 5 geometry)",
            "allgather_us": round(us, 1), "shard_bytes": shard,
            "GBs_per_link": round(shard / (us * 1e-6) / 1e9, 1), "link_peak_GBs": 153.0,
            "received_GBs_per_gpu": round((world - 1) * shard / (us * 1e-6) / 1e9, 1),
            "bank_us": round(float(t[1]) * 1e3, 1)}


def launch_command(argv, gpus, port=None):
    """The command that starts ``bench.py`` as ``gpus`` ranks of one node, one per GPU over RCCL: what the driver
    runs for N > 1, and what ``python bench.py --gpus N`` re-executes itself as when nobody did
    (reference bootstrap: mm_interleaved/utils/misc.py:292-337).  Rendezvous on 127.0.0.1 (the container's
    hostname may not resolve)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(argv, gpus):
    """--gpus N > 1 without a launcher around it: fail loudly when the node has fewer GPUs, else re-execute under
    torch.distributed.run and return its exit status.  Never degrades to one rank."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < gpus:
        print(f"bench.py: --gpus {gpus} but this node shows {have} GPU(s); refusing to run fewer ranks than asked",
              file=sys.stderr, flush=True)
        return 2
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    return subprocess.run(launch_command(argv, gpus), env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2_northstar", choices=sorted(WORKLOADS))
    ap.add_argument("--nq", type=int, default=None, help="override Nq (parity/sweep use)")
    ap.add_argument("--dtype", default=None, choices=sorted(DTYPES))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="diagnostic: no per-kernel event pass; prints ms/step only")
    ap.add_argument("--event-steps", type=int, default=10, help="steps of the per-kernel event pass (before the warm-up)")
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="untimed steps for this long before anything is measured: a GPU coming out of idle runs its "
                         "first ~30-50 ms of work 7 %% slower (profiles/r02s_clock_ramp.log), which is the whole "
                         "timed region of a --steps 20 run")
    ap.add_argument("--visible", default="all", choices=["all", "causal"],
                    help="causal: image k of n is visible to the queries after k/n of the sequence, zero attention elsewhere")
    ap.add_argument("--loc-dist", default="uniform", choices=["uniform", "centre"],
                    help="sampling locations: uniform over each level (the contract workload) or clustered "
                         "around one reference point (what the LLM path produces)")
    ap.add_argument("--fresh-levels", action="store_true",
                    help="the literal drop-in: spatial_shapes / level_start_index rebuilt as NEW device tensors on every "
                         "call, as the reference's callers do (modeling_llama_mmfs.py:298-308, sd_mmfs.py:31-41) -- the shim "
                         "has never seen them, the backward checks the table on the device and has no hybrid routing")
    ap.add_argument("--fresh-levels-unused", action="store_true",
                    help="measurement: build the two fresh level tensors per call as --fresh-levels does, but hand the op the "
                         "registered pair -- what of the fresh line's distance to the default line is the CALLER's tensor construction")
    ap.add_argument("--grad", default="randn", choices=["randn", "ones"],
                    help="grad_output: N(0,1) or ones (the reference's speed test backpropagates .sum())")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the two sibling measurements of the default line (N = 1): the fp16 run of the same shape with its "
                         "error against the oracle slab, and the reference's call pattern (fresh level tensors per call)")
    ap.add_argument("--exchange-in-step", action="store_true",
                    help="N > 1: the feature all-gather of BASELINE config 5 is issued on a side stream inside every "
                         "timed step, overlapping the op (SURVEY 8e); default: timed after the main region")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                 f"--nproc-per-node {args.gpus}, or unset WORLD_SIZE and let bench.py launch its own ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback exists)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    import MultiScaleDeformableAttention as MSDA
    from mmfs_amd.functions import MSDeformAttnFunction

    w = dict(WORKLOADS[args.workload])
    if args.nq:
        w["Nq"] = args.nq
    if args.dtype:
        w["dtype"] = args.dtype
    value, shapes, start, loc, attn, grad = make_inputs(w, device, seed=rank, loc_dist=args.loc_dist, visible=args.visible)
    # the level tables through the product's own constructor (mmfs_amd.levels.make_level_tables, what the MMFS
    # blocks use): built once, cached, and known to the shim without a device->host copy -- the reference's
    # callers rebuild both tensors on every call (tables the shim has never seen are checked on the device)
    from mmfs_amd.levels import make_level_tables
    shapes, start, _ = make_level_tables(w["shapes"], w["n"], device)
    value.requires_grad_(True); loc.requires_grad_(True); attn.requires_grad_(True)
    if args.grad == "ones":
        grad = torch.ones_like(grad)
    host_shapes = torch.tensor(w["shapes"] * w["n"], dtype=torch.long)

    overlap = None
    if args.exchange_in_step and dist is not None:
        overlap = ExchangeOverlap(device, rank, world)

    def step(fresh=False):
        if overlap is not None:
            overlap.issue()                     # the all-gather rides a side stream under the op
        if args.fresh_levels or args.fresh_levels_unused or fresh:
            # what the reference's callers do per call: two new device tensors, nothing registered
            sh = host_shapes.to(device)
            st = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
            if args.fresh_levels_unused:      # (measurement: the caller's tensor construction alone -- the op gets the registered pair)
                sh, st = shapes, start
            out = MSDeformAttnFunction.apply(value, sh, st, loc, attn, 1)
        else:
            out = MSDeformAttnFunction.apply(value, shapes, start, loc, attn, 1)
        res = torch.autograd.grad(out, (value, loc, attn), grad)
        if overlap is not None:
            overlap.join()
        return res

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- parity guard: no number is printed for kernels that compute something else.  Size-independent
    # properties of the op on this very workload (tests/test_op_gpu.py::test_full_size_properties): out is
    # linear in value and in attn, so <value, dL/dvalue> = <attn, dL/dattn> = <out, grad> (Euler); finite outputs.
    # (MMFS_EXPERIMENT=1: kernel-timing experiments with deliberately wrong arithmetic; the line is tagged)
    experiment = os.environ.get("MMFS_EXPERIMENT") == "1"
    out = MSDeformAttnFunction.apply(value, shapes, start, loc, attn, 1)
    gv, gl, ga = torch.autograd.grad(out, (value, loc, attn), grad)
    og = (out.double() * grad.reshape(out.shape).double()).sum()
    rel = {"f32": 1e-4, "f16": 4e-3, "bf16": 3e-2}[w["dtype"]]
    e1 = abs(float((gv.double() * value.double()).sum() - og)); e2 = abs(float((ga.double() * attn.double()).sum() - og))
    assert experiment or bool(torch.isfinite(out).all() and torch.isfinite(gv).all() and torch.isfinite(gl).all() and torch.isfinite(ga).all()), "non-finite output"
    assert experiment or e1 <= rel * abs(float(og)) + rel and e2 <= rel * abs(float(og)) + rel, f"parity guard failed: {e1:.3e} {e2:.3e} vs {float(og):.3e}"
    del out, gv, gl, ga

    # ---- per-kernel HIP events, recorded on the launch stream around every C-ABI launch: a pass of its
    # own, BEFORE the warm-up (bracketing every launch costs ~45 us of host work per step, and the stage-by-
    # stage calls are not what production issues: neither belongs in the timed region)
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    log = []
    if not args.no_kernel_events:
        for _ in range(max(args.warmup, 40)):      # (its own warm-up: the averages are of a warm GPU, like the rocprofv3 trace's)
            step()
        MSDA._event_log = log
        for i in range(max(1, args.event_steps)):
            step()
        fence()
        MSDA._event_log = None
    for _ in range(args.warmup):
        step()
    fence()
    # ---- the timed region: K clean steps (every pass ONE C call, like production)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if args.no_kernel_events:
        if rank == 0:
            print(json.dumps({"ms_per_step": round(elapsed / args.steps * 1e3, 4), "note": "no kernel events"}))
        return

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        e = torch.empty((), dtype=DTYPES[w["dtype"]]).element_size()
        per_kernel = {}
        for name, a, b in log:
            per_kernel.setdefault(name, []).append(a.elapsed_time(b))     # ms
        mean_ms = {k: sum(v) / len(v) for k, v in per_kernel.items()}
        ab = algorithmic_bytes(w, e)
        if "msda_bwd_taps_coarse" in mean_ms:          # the gather kernel then covers the other levels only
            ab["msda_bwd_taps"] = ab["msda_bwd_taps_fine"]
        dom = max((k for k in mean_ms if k in ab), key=lambda k: mean_ms[k])
        achieved = ab[dom] / (mean_ms[dom] * 1e-3) / 1e9
        traffic = traffic_from = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")          # from a separate rocprofv3 --pmc run
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                traffic = j.get(args.workload, {}).get(dom)
                prov = j.get("_source")
                if isinstance(prov, dict):
                    prov = prov.get(args.workload)
                traffic_from = prov if traffic is not None else None
            except Exception:
                traffic = None
        Leff = len(w["shapes"]) * w["n"]
        res = {
            "metric": "mmfs_ms_deform_attn_fwd_bwd_samples_per_sec",
            "value": round(world * w["B"] * args.steps / elapsed, 2),
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": w["dtype"], "data": "synthetic" if not experiment else "synthetic; EXPERIMENT BUILD, PARITY GUARD OFF -- not a result",
            "config": {"workload": f"{args.workload}{'' if args.loc_dist == 'uniform' else '@' + args.loc_dist}{'' if args.visible == 'all' else '@' + args.visible}: ms_deform_attn fwd+bwd, per-GPU B={w['B']} Nq={w['Nq']} "
                                   f"L={Leff} H={w['H']} P={w['P']} C={w['H'] * w['D']} S={sum(h * x for h, x in w['shapes']) * w['n']}",
                       "global_batch": world * w["B"], "parallelism": f"batch-sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_from": traffic_from,
                         "algorithmic_bytes": ab[dom], "mean_us": round(mean_ms[dom] * 1e3, 2)},
            "kernels_mean_us": {k: round(v * 1e3, 2) for k, v in mean_ms.items()},
            # every kernel of the step against the same roofline (algorithmic bytes of ITS tensors / its mean time / peak):
            # with three kernels within a few us of each other, which one is "dominant" changes from box to box
            "kernels_frac": {k: round(ab[k] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for k, v in mean_ms.items()
                             if k in ab and v > 0 and ab[k] > 0},
            # event brackets cost a few us each and span a stage's helper launches: their sum may exceed the clean step
            "event_overhead_us": round(sum(mean_ms.values()) * 1e3 - elapsed / args.steps * 1e6, 2),
            "levels": "fresh per call, unregistered (reference call pattern)" if args.fresh_levels else
                      "fresh tensors built per call but NOT used (registered pair to the op)" if args.fresh_levels_unused else
                      "make_level_tables (built once, known to the shim)",
            # whole step against the roofline: algorithmic bytes of forward + backward / the clean step time
            "fwdbwd_hbm_frac": round(ab["fwdbwd"] / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
            "kernels_hbm_frac": round(ab["fwdbwd"] / (sum(mean_ms.values()) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        }
        # the WHOLE step against the roofline (SURVEY 8d's fwdbwd_bytes / the clean step time), against the 8 TB/s of the
        # data sheet and against the 6.29 TB/s a plain copy reaches on this part (MI355X_MICROARCH.md); forward and
        # backward separately from the per-kernel events
        step_s = elapsed / args.steps
        bwd_us = sum(v for k, v in mean_ms.items() if k != "msda_fwd") * 1e3
        res["step_roofline"] = {
            "bound": "hbm", "bytes": ab["fwdbwd"], "ms": round(step_s * 1e3, 4),
            "achieved": round(ab["fwdbwd"] / step_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ab["fwdbwd"] / step_s / 1e9 / HBM_PEAK_GBS, 4),
            "copy_ceiling": HBM_COPY_GBS, "frac_of_copy_ceiling": round(ab["fwdbwd"] / step_s / 1e9 / HBM_COPY_GBS, 4),
            "forward": {"bytes": ab["msda_fwd"], "us": round(mean_ms.get("msda_fwd", 0.0) * 1e3, 2),
                        "frac": round(ab["msda_fwd"] / (mean_ms["msda_fwd"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if mean_ms.get("msda_fwd") else None},
            "backward": {"bytes": ab["bwd"], "us": round(bwd_us, 2),
                         "frac": round(ab["bwd"] / (bwd_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if bwd_us > 0 else None},
        }
        if world == 1 and not args.no_cpu_baseline and not args.no_extras and args.workload == "cfg2_northstar" \
                and not (args.fresh_levels or args.fresh_levels_unused) and not experiment:
            # the reference's call pattern on the same tensors: spatial_shapes / level_start_index rebuilt per call as its
            # callers do (modeling_llama_mmfs.py:298-308) -- what an UNCHANGED reference gets from the drop-in
            # (siblings of the headline: a failure in one of them must not cost the line its headline)
            try:
                res["dropin_unchanged_ms"] = round(timed_steps(lambda: step(fresh=True), 20, 10), 4)
            except Exception as e:      # noqa: BLE001
                res["dropin_unchanged_ms"] = None
                res["dropin_unchanged_error"] = f"{type(e).__name__}: {e}"
            if w["dtype"] != "f16":
                try:
                    res["fp16"] = fp16_sibling(w, device, shapes, start)
                except Exception as e:  # noqa: BLE001
                    res["fp16"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:      # noqa: BLE001  (the checker is test infrastructure: its absence is reported, not fatal)
                res["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    ex = None
    if dist is not None:
        # every rank says whether it can enter the collectives BEFORE any of them does: a rank that fails
        # inside its own set-up must not leave the others waiting in an all-gather
        ok = torch.ones((), dtype=torch.int32, device=device)
        try:
            from mmfs_amd import bank as _bank      # noqa: F401
        except Exception as e:
            ok.zero_()
            ex = {"error": f"{type(e).__name__}: {e}"}
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            ex = exchange_step(device, rank, world, dist)          # a failure inside a collective propagates: no hang
            if overlap is not None:
                ex["in_step"] = "all-gather issued on a side stream in every timed step (its time is inside ms_per_step)"
        elif ex is None:
            ex = {"error": "another rank could not set the exchange up"}
    if rank == 0:
        if ex is not None:
            res["exchange"] = ex
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
