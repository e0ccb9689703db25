#!/usr/bin/env python3
"""Host floor of one op step: where the time of a SMALL forward + backward goes.

For each shape: wall time per step (autograd forward + backward through the shim, synchronised only
around the whole loop), the GPU-busy time of the same step (sum of the kernels, from the per-kernel
event log), the CPU time to ISSUE a step (loop without the final synchronise, GPU allowed to lag), and
a cProfile of the issue path.  usage: python tools/host_floor.py [--profile]
"""
import cProfile
import io
import json
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch  # noqa: E402
import MultiScaleDeformableAttention as MSDA  # noqa: E402
from mmfs_amd.functions import MSDeformAttnFunction  # noqa: E402
from mmfs_amd.levels import make_level_tables  # noqa: E402

SHAPES = {
    "cfg1 (B=2 Nq=1024 H=8 D=32 P=4 fp32)": dict(B=2, Nq=1024, H=8, D=32, P=4, dt=torch.float32),
    "Nq=64 (B=2 H=8 D=32 P=4 fp32)": dict(B=2, Nq=64, H=8, D=32, P=4, dt=torch.float32),
    "decode (B=4 Nq=1 H=16 D=64 P=8 bf16, 3 levels)": dict(B=4, Nq=1, H=16, D=64, P=8, dt=torch.bfloat16,
                                                          shapes=[(32, 32), (16, 16), (8, 8)]),
}


def make(w, dev="cuda"):
    shapes = w.get("shapes", [(64, 64), (32, 32), (16, 16), (8, 8)])
    sh, st, _ = make_level_tables(shapes, 1, dev)
    S = sum(h * ww for h, ww in shapes)
    g = torch.Generator(device=dev).manual_seed(0)
    B, Nq, H, D, P, dt = w["B"], w["Nq"], w["H"], w["D"], w["P"], w["dt"]
    L = len(shapes)
    value = torch.rand(B, S, H, D, device=dev, generator=g).to(dt).requires_grad_(True)
    loc = torch.rand(B, Nq, H, L, P, 2, device=dev, generator=g).to(dt).requires_grad_(True)
    attn = torch.rand(B, Nq, H, L, P, device=dev, generator=g)
    attn = (attn / attn.sum((-1, -2), keepdim=True)).to(dt).requires_grad_(True)
    grad = torch.randn(B, Nq, H * D, device=dev, generator=g).to(dt)
    return value, sh, st, loc, attn, grad


def main():
    prof = "--profile" in sys.argv
    for name, w in SHAPES.items():
        value, sh, st, loc, attn, grad = make(w)

        def step():
            out = MSDeformAttnFunction.apply(value, sh, st, loc, attn, 64)
            torch.autograd.grad(out, (value, loc, attn), grad)

        for _ in range(20):
            step()
        torch.cuda.synchronize()
        # the host of a GPU box is shared: best and median of 7 runs of 200 steps
        n, issue, wall = 200, [], []
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            issue.append((time.perf_counter() - t0) / n)
            torch.cuda.synchronize()
            wall.append((time.perf_counter() - t0) / n)
        t_issue, t_wall = min(issue), min(wall)
        wall_med = sorted(wall)[len(wall) // 2]
        # GPU-busy time: per-kernel events (stage-by-stage calls: more host work, same kernels)
        log = []
        MSDA._event_log = log
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        MSDA._event_log = None
        gpu_us = sum(a.elapsed_time(b) for _, a, b in log) * 1e3 / 10
        names = {}
        for k, a, b in log:
            names[k] = names.get(k, 0.0) + a.elapsed_time(b) * 1e3 / 10
        print(json.dumps({"shape": name, "wall_us_per_step": round(t_wall * 1e6, 1), "wall_us_median": round(wall_med * 1e6, 1),
                          "issue_us_per_step": round(t_issue * 1e6, 1), "gpu_busy_us": round(gpu_us, 1),
                          "kernels_us": {k: round(v, 1) for k, v in names.items()}}), flush=True)
        if prof:
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(200):
                step()
            pr.disable()
            torch.cuda.synchronize()
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
            print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:3500], flush=True)


if __name__ == "__main__":
    main()
