#!/usr/bin/env python3
"""How long does the GPU take to reach its working clocks?  One forward + backward step of the north-star
workload in a loop from an idle GPU; mean step time per window of 50 steps (HIP events), and the forward
alone the same way.  usage: python tools/clock_ramp.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
import MultiScaleDeformableAttention as MSDA
from mmfs_amd.functions import MSDeformAttnFunction
from mmfs_amd.levels import make_level_tables
import bench

w = bench.WORKLOADS["cfg2_northstar"]
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
B, Nq, H, D, P = w["B"], w["Nq"], w["H"], w["D"], w["P"]
sh, st, S = make_level_tables(w["shapes"], 1, dev)
L = sh.shape[0]
value = torch.rand(B, S, H, D, device=dev, generator=g).bfloat16().requires_grad_(True)
loc = torch.rand(B, Nq, H, L, P, 2, device=dev, generator=g).bfloat16().requires_grad_(True)
attn = torch.rand(B, Nq, H, L, P, device=dev, generator=g)
attn = (attn / attn.sum((-1, -2), keepdim=True)).bfloat16().requires_grad_(True)
grad = torch.randn(B, Nq, H * D, device=dev, generator=g).bfloat16()


def step():
    out = MSDeformAttnFunction.apply(value, sh, st, loc, attn, 1)
    torch.autograd.grad(out, (value, loc, attn), grad)


step(); torch.cuda.synchronize()
time.sleep(3.0)                                      # idle GPU
ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
ev[0].record()
for i in range(60):
    for _ in range(50):
        step()
    ev[i + 1].record()
torch.cuda.synchronize()
t = 0.0
line = []
for i in range(60):
    ms = ev[i].elapsed_time(ev[i + 1])
    t += ms
    line.append("%.0fms:%.3f" % (t, ms / 50))
print("step ms (window of 50 steps, time since start): " + "  ".join(line[:8]) + "  ...  " + "  ".join(line[-4:]))


def series(fn, label, windows=40, per=100):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(windows + 1)]
    ev[0].record()
    for i in range(windows):
        for _ in range(per):
            fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    t, line = 0.0, []
    for i in range(windows):
        ms = ev[i].elapsed_time(ev[i + 1]); t += ms
        line.append("%.0fms:%.1f" % (t, ms / per * 1e3))
    print(label + " us per call (window of %d, time since start): " % per + "  ".join(line[:6]) + "  ...  " + "  ".join(line[-3:]))


vd, ld, ad = value.detach(), loc.detach(), attn.detach()
fwd = lambda: MSDA.ms_deform_attn_forward(vd, sh, st, ld, ad, 1)
series(fwd, "forward alone, right after the step loop,")
time.sleep(3.0)
series(fwd, "forward alone, from idle,")
vz = torch.full_like(vd, 0.5)
series(lambda: MSDA.ms_deform_attn_forward(vz, sh, st, ld, ad, 1), "forward alone, value = 0.5 everywhere,", windows=10)
series(step, "whole step again,", windows=6, per=50)
