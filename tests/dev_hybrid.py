"""Development check of the hybrid (dense small levels) path against the plain kernels and the
oracle, plus per-kernel timings.  Run on the GPU box:  python tests/dev_hybrid.py [--quick]"""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mm-interleaved_amd"))
import MultiScaleDeformableAttention as MSDA  # noqa: E402

DT = {"bf16": torch.bfloat16, "f16": torch.float16}


def make(B, Nq, H, D, P, shapes, n, dtype, seed=0, dev="cuda"):
    g = torch.Generator(device=dev).manual_seed(seed)
    sh = torch.tensor(shapes * n, dtype=torch.long, device=dev)
    st = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    S, L = int(sh.prod(1).sum()), sh.shape[0]
    value = (torch.rand(B, S, H, D, device=dev, generator=g) - 0.5).to(dtype)
    loc = (torch.rand(B, Nq, H, L, P, 2, device=dev, generator=g) * 1.2 - 0.1).to(dtype)
    attn = torch.rand(B, Nq, H, L, P, device=dev, generator=g) + 1e-5
    attn = (attn / attn.sum((-1, -2), keepdim=True)).to(dtype)
    go = (torch.rand(B, Nq, H * D, device=dev, generator=g) - 0.5).to(dtype)
    return value, sh, st, loc, attn, go, S


def run(args, hybrid, log=None):
    value, sh, st, loc, attn, go, S = args
    MSDA._hybrid = hybrid
    MSDA._event_log = log
    out = MSDA.ms_deform_attn_forward(value, sh, st, loc, attn, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(value, sh, st, loc, attn, go, 64)
    MSDA._event_log = None
    torch.cuda.synchronize()
    return out, gv, gl, ga


def cmp(name, a, b):
    a, b = a.float(), b.float()
    d = (a - b).abs()
    bad = ~torch.isfinite(a) | ~torch.isfinite(b)
    print(f"    {name:11s} max|d|={d[~bad].max().item():.3e}  mean|d|={d[~bad].mean().item():.3e}  "
          f"max|ref|={b[~bad].abs().max().item():.3e}  nonfinite={int(bad.sum())}", flush=True)


def oracle_cmp(args, res, tag):
    from oracle import msda_oracle
    value, sh, st, loc, attn, go, S = args
    f = lambda t: t.detach().float().cpu().numpy().astype(np.float64)
    o = msda_oracle.forward(f(value), sh.cpu().numpy(), st.cpu().numpy(), f(loc), f(attn))
    gvo, glo, gao = msda_oracle.backward(f(value), sh.cpu().numpy(), st.cpu().numpy(), f(loc), f(attn), f(go))
    for n, a, b in (("out", res[0], o), ("grad_value", res[1], gvo), ("grad_loc", res[2], glo), ("grad_attn", res[3], gao)):
        a = a.detach().float().cpu().numpy().astype(np.float64).reshape(b.shape)
        print(f"    oracle[{tag}] {n:11s} max|d|={np.abs(a - b).max():.3e}  max|ref|={np.abs(b).max():.3e}", flush=True)


def timing(args, hybrid, iters=20):
    for _ in range(3):
        run(args, hybrid)
    log = []
    for _ in range(iters):
        run(args, hybrid, log)
    torch.cuda.synchronize()
    acc = {}
    for name, t0, t1 in log:
        acc.setdefault(name, []).append(t0.elapsed_time(t1) * 1e3)
    tot = 0.0
    for k, v in acc.items():
        m = sum(v) / len(v)
        tot += m
        print(f"    {k:26s} {m:9.1f} us", flush=True)
    print(f"    {'sum':26s} {tot:9.1f} us", flush=True)
    t = time.perf_counter()
    for _ in range(iters):
        run(args, hybrid)
    torch.cuda.synchronize()
    print(f"    wall per fwd+bwd            {(time.perf_counter() - t) / iters * 1e6:9.1f} us", flush=True)
    if hybrid and hasattr(MSDA._lib, "mmfs_debug_dense_profile"):
        import ctypes
        buf = (ctypes.c_ulonglong * 32)()
        MSDA._lib.mmfs_debug_dense_profile(buf, 1)
        run(args, True)
        MSDA._lib.mmfs_debug_dense_profile(buf, 1)
        v = list(buf)
        print("    taps phases (kclk summed over WGs): mfma %d | reload+G %d | bar1 %d | lookup %d | bar2 %d" %
              tuple(x // 1000 for x in v[0:5]), flush=True)
        for lv in (0, 1):
            o = 8 + 8 * lv
            print("    value L%d phases (kclk): g+bar %d | zero %d | record %d | apply %d | split %d | mfma %d" %
                  ((lv,) + tuple(x // 1000 for x in v[o:o + 6])), flush=True)


CASES = [
    # name, B, Nq, H, D, P, shapes, n, dtype, oracle?
    ("small_bf16", 2, 200, 4, 128, 4, [(16, 16), (8, 8), (20, 20), (5, 7)], 1, "bf16", True),
    ("small_f16_d64", 2, 333, 3, 64, 8, [(32, 32), (16, 16), (8, 8)], 2, "f16", True),
    ("small_bf16_d32", 1, 130, 2, 32, 4, [(16, 16), (3, 3)], 1, "bf16", True),
    ("northstar", 8, 4096, 8, 128, 4, [(64, 64), (32, 32), (16, 16), (8, 8)], 1, "bf16", False),
    ("sd_real", 8, 4096, 16, 64, 8, [(64, 64), (32, 32), (16, 16), (8, 8)], 1, "bf16", False),
    ("llm_n4", 4, 2048, 16, 64, 8, [(32, 32), (16, 16), (8, 8)], 4, "bf16", False),
]

if __name__ == "__main__":
    quick = "--quick" in sys.argv
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    for name, B, Nq, H, D, P, shapes, n, dt, orc in CASES:
        if only and name not in only:
            continue
        print(f"== {name}: B={B} Nq={Nq} H={H} D={D} P={P} levels={shapes}x{n} {dt}", flush=True)
        try:
            args = make(B, Nq, H, D, P, shapes, n, DT[dt])
            MSDA.register_level_tables(args[1], args[2], args[6])
            plain = run(args, False)
            hyb = run(args, True)
            for nm, a, b in zip(("out", "grad_value", "grad_loc", "grad_attn"), hyb, plain):
                cmp(nm, a, b)
            if orc:
                oracle_cmp(args, plain, "plain")
                oracle_cmp(args, hyb, "hybrid")
            if not quick and not orc:
                print("  plain:", flush=True)
                timing(args, False)
                print("  hybrid:", flush=True)
                timing(args, True)
        except Exception:
            traceback.print_exc()
