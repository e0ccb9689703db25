#!/usr/bin/env python3
"""Backward of the LLM geometry (4 images, causal visibility: bench.py --visible causal) with and
without MMFS_BWD_LAZY_ZERO_ATTN, per-kernel HIP-event times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch  # noqa: E402
import bench  # noqa: E402
import MultiScaleDeformableAttention as MSDA  # noqa: E402
w = dict(bench.WORKLOADS["cfg5_llm_n4"])
value, shapes, start, loc, attn, grad = bench.make_inputs(w, "cuda", 0, visible="causal")
for lazy in (False, True, False, True):
    MSDA._event_log = log = []
    for _ in range(30):
        MSDA.ms_deform_attn_backward(value, shapes, start, loc, attn, grad, 1, lazy_zero_attn=lazy)
    torch.cuda.synchronize(); MSDA._event_log = None
    ev = {}
    for n, a, b in log[len(log)//3:]:
        ev.setdefault(n, []).append(a.elapsed_time(b) * 1e3)
    print("lazy" if lazy else "full", {k: round(sum(v)/len(v), 1) for k, v in ev.items()})
