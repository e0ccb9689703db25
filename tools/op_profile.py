#!/usr/bin/env python3
"""Kernel-level breakdown (torch profiler) of one bench.py workload: python tools/op_profile.py cfg2_northstar [centre]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
import bench
from mmfs_amd.functions import MSDeformAttnFunction
from torch.profiler import profile, ProfilerActivity

w = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2_northstar"])
dist = sys.argv[2] if len(sys.argv) > 2 else "uniform"
value, shapes, start, loc, attn, grad = bench.make_inputs(w, torch.device("cuda"), 0, dist)
value.requires_grad_(True); loc.requires_grad_(True); attn.requires_grad_(True)
def step():
    out = MSDeformAttnFunction.apply(value, shapes, start, loc, attn, 1)
    torch.autograd.grad(out, (value, loc, attn), grad)
for _ in range(5): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(10): step()
    torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:12]:
    print(f"{e.device_time_total / 10:9.1f} us  x{e.count // 10:<3d} {e.key[:100]}")
