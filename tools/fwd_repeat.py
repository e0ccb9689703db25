#!/usr/bin/env python3
"""The forward alone, launched back to back (what a kernel micro-benchmark does), against the same
kernel inside a forward + backward step: does the context (what the backward leaves in L2 / MALL / TLBs)
matter?  usage: python tools/fwd_repeat.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
import MultiScaleDeformableAttention as MSDA
from mmfs_amd.levels import make_level_tables
import bench

w = bench.WORKLOADS["cfg2_northstar"]
value, _, _, loc, attn, grad = bench.make_inputs(w, "cuda", 0)[:6] if False else (None,) * 6
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
B, Nq, H, D, P = w["B"], w["Nq"], w["H"], w["D"], w["P"]
sh, st, S = make_level_tables(w["shapes"], 1, dev)
L = sh.shape[0]
value = torch.rand(B, S, H, D, device=dev, generator=g).bfloat16()
loc = torch.rand(B, Nq, H, L, P, 2, device=dev, generator=g).bfloat16()
attn = torch.rand(B, Nq, H, L, P, device=dev, generator=g)
attn = (attn / attn.sum((-1, -2), keepdim=True)).bfloat16()
grad = torch.randn(B, Nq, H * D, device=dev, generator=g).bfloat16()


def timed(fn, n):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fwd = lambda: MSDA.ms_deform_attn_forward(value, sh, st, loc, attn, 1)
print("forward back to back: %.1f us" % timed(fwd, 50))
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
def fwd_cold():
    big.zero_()
    return fwd()
t_all = timed(fwd_cold, 20); t_z = timed(lambda: big.zero_(), 20)
print("forward after a 1 GiB memset each time: %.1f us (memset alone %.1f)" % (t_all - t_z, t_z))
loc1 = loc * 0 + 0.5
print("forward, every sample at the map centre: %.1f us" % timed(lambda: MSDA.ms_deform_attn_forward(value, sh, st, loc1, attn, 1), 50))
loc2 = (torch.rand(B, Nq, H, L, P, 2, device=dev, generator=g) * 0.96 + 0.02).bfloat16()
print("forward, samples kept off the borders: %.1f us" % timed(lambda: MSDA.ms_deform_attn_forward(value, sh, st, loc2, attn, 1), 50))
# data-dependent power?  the same kernel, the same addresses, other VALUES
vz = torch.zeros_like(value)
print("forward, value all zeros: %.1f us" % timed(lambda: MSDA.ms_deform_attn_forward(vz, sh, st, loc, attn, 1), 50))
vc = torch.full_like(value, 0.5)
print("forward, value all 0.5: %.1f us" % timed(lambda: MSDA.ms_deform_attn_forward(vc, sh, st, loc, attn, 1), 50))
print("forward, random value again: %.1f us" % timed(fwd, 50))
