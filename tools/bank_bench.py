#!/usr/bin/env python3
"""Feature-bank construction: the one-pass kernel (csrc/mmfs_bank.hip) against the
framework-op statement (concatenate transposed views, index_select, mask), forward and backward.
HBM-bound copy: bytes = 2 * n_slots * S * C * e;  GB/s against the 8 TB/s peak."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch  # noqa: E402
from mmfs_amd.functions import BankGatherFunction  # noqa: E402

dev, dt = "cuda", torch.bfloat16
CASES = {
    "llm  B=4 n=4 (32,16,8)^2 C=1024": (16, [32, 16, 8], [0, 1, 2, 3, 4, 5, -1, -1, 6, 7, 8, 9, 10, 11, 12, -1]),
    "sd   B=8 n=1 (64,32,16,8)^2 C=1024": (8, [64, 32, 16, 8], list(range(8))),
    "llm  B=32 n=2 (32,16,8)^2 C=1024": (64, [32, 16, 8], list(range(64))),
}


def torch_bank(levels, src):
    packed = torch.cat([f.flatten(2).transpose(1, 2) for f in levels], dim=1)
    valid = (src >= 0) & (src < packed.shape[0])
    return packed.index_select(0, src.clamp(0, packed.shape[0] - 1)) * valid[:, None, None].to(packed.dtype)


def timed(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e6


for name, (n_img, sides, slots) in CASES.items():
    levels = [torch.randn(n_img, 1024, s, s, device=dev, dtype=dt) for s in sides]
    src = torch.tensor(slots, device=dev)
    S = sum(s * s for s in sides)
    nbytes = 2 * len(slots) * S * 1024 * 2
    assert torch.equal(BankGatherFunction.apply(src, *levels), torch_bank(levels, src))
    t_k = timed(lambda: BankGatherFunction.apply(src, *levels))
    t_t = timed(lambda: torch_bank(levels, src))
    lv = [f.clone().requires_grad_(True) for f in levels]
    g = torch.randn(len(slots), S, 1024, device=dev, dtype=dt)

    def fb(fn):
        for f in lv:
            f.grad = None
        fn(lv, src).backward(g)
    t_kb = timed(lambda: fb(lambda l, s: BankGatherFunction.apply(s, *l)))
    t_tb = timed(lambda: fb(torch_bank))
    import MultiScaleDeformableAttention as MSDA
    MSDA._event_log = log = []
    for _ in range(20):
        fb(lambda l, s: BankGatherFunction.apply(s, *l))
    torch.cuda.synchronize()
    MSDA._event_log = None
    ev = {}
    for n_, a, b in log[10:]:
        ev.setdefault(n_, []).append(a.elapsed_time(b) * 1e3)
    t_sc = sum(ev["mmfs_bank_scatter"]) / len(ev["mmfs_bank_scatter"])
    print(f"{name:40s} scatter kernel alone {t_sc:7.1f} us = {nbytes / t_sc / 1e3:6.0f} GB/s")
    print(f"{name:40s} bank {nbytes / 2 / 2**20:6.1f} MiB | kernel {t_k:7.1f} us = {nbytes / t_k / 1e3:6.0f} GB/s "
          f"({nbytes / t_k / 1e3 / 8000:.0%} of HBM peak) | framework ops {t_t:7.1f} us | fwd+bwd {t_kb:7.1f} vs {t_tb:7.1f} us")
