#!/usr/bin/env python3
"""Why does the forward take 175 us inside a forward + backward step and 148 us alone (tools/clock_ramp.py)?
Steady-state timings of the forward kernel (HIP events around its launch only) in four contexts:
  alone, back to back | after a 512 MiB memset (caches polluted, little power) | after the backward (the step)
  | after the backward and 1 ms of idle spinning (caches as in the step, power budget recovered)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
import MultiScaleDeformableAttention as MSDA
from mmfs_amd.levels import make_level_tables
import bench

w = bench.WORKLOADS["cfg2_northstar"]
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
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)

fwd = lambda: MSDA.ms_deform_attn_forward(value, sh, st, loc, attn, 1)
bwd = lambda: MSDA.ms_deform_attn_backward(value, sh, st, loc, attn, grad, 1)


def fwd_time(before, n=300, warm=300):
    for _ in range(warm):
        before(); fwd()
    evs = []
    for _ in range(n):
        before()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fwd(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return ts[len(ts) // 2]


print("forward alone, back to back:                 %.1f us" % fwd_time(lambda: None))
print("forward after a 512 MiB memset:              %.1f us" % fwd_time(lambda: big.zero_()))
print("forward after the backward (= the step):     %.1f us" % fwd_time(bwd))
print("forward after the backward + 1 ms spinning:  %.1f us" % fwd_time(lambda: (bwd(), torch.cuda._sleep(2_000_000))))
print("forward after 1 ms spinning only:            %.1f us" % fwd_time(lambda: torch.cuda._sleep(2_000_000)))
# which tensor has to be cache-resident?  pollute everything, then re-touch one of them before the forward
touch = lambda t: t.view(torch.int16).sum()
print("memset, then value re-read:                  %.1f us" % fwd_time(lambda: (big.zero_(), touch(value))))
print("memset, then loc + attn re-read:             %.1f us" % fwd_time(lambda: (big.zero_(), touch(loc), touch(attn))))
print("memset, then value + loc + attn re-read:     %.1f us" % fwd_time(lambda: (big.zero_(), touch(value), touch(loc), touch(attn))))
small = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
print("forward after a 64 MiB memset:               %.1f us" % fwd_time(lambda: small.zero_()))
