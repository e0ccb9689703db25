"""Shared test helpers: seeded inputs in the reference's test distribution
(ops/tests/create_data.py:11-30) and golden loading."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def level_tables(shapes, device="cpu"):
    sh = torch.as_tensor(shapes, dtype=torch.long, device=device).reshape(-1, 2)
    start = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    return sh.contiguous(), start.contiguous()


def make_inputs(B, H, D, Nq, P, shapes, seed=0, loc_range=(0.0, 1.0), dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    sh, start = level_tables(shapes)
    S, L = int(sh.prod(1).sum()), sh.shape[0]
    value = torch.rand(B, S, H, D, generator=g)
    lo, hi = loc_range
    loc = torch.rand(B, Nq, H, L, P, 2, generator=g) * (hi - lo) + lo
    attn = torch.rand(B, Nq, H, L, P, generator=g) + 1e-5
    attn = attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    grad = torch.randn(B, Nq, H * D, generator=g)
    # round through the storage dtype so oracle and device see identical numbers
    rt = lambda t: t.to(dtype).to(torch.float64)
    return dict(value=rt(value), shapes=sh, start=start, loc=rt(loc), attn=rt(attn), grad=rt(grad))


def max_abs(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64).reshape(a.shape)
    return float(np.max(np.abs(a - b))) if a.size else 0.0
