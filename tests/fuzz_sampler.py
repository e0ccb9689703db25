"""Randomised bit-identity sweep: the fused sampler (plan -> sampler in one kernel, mmfs_sample_fwd) against the
two-kernel path (plan kernel + op) on random small MMFS modules, three storage types.  Not collected by pytest
(a GPU box runs it on demand):

    python tests/fuzz_sampler.py [n_cases] [seed]

Half of the cases use "nice" numbers -- weights and queries that are multiples of small powers of two -- so that the
plan's fp32 results land exactly on 16-bit rounding ties, where a fused multiply-add that rounds once and a
multiply-add that rounds to fp32 first part (profiles/r03_experiments.md r03bp: fp16, v_fma_mixlo_f16).
"""
import contextlib, io, os, random, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch

from mmfs_amd.levels import make_level_tables
from mmfs_amd.modules import MMFS

DEV = "cuda"


def one_case(rng, idx):
    H = rng.choice([2, 4, 8])
    D = rng.choice([8, 16, 32, 64])
    L = rng.choice([1, 2, 3, 4])
    P = rng.choice([4, 8])
    n = rng.choice([1, 2, 3])
    sizes = [rng.choice([1, 2, 4, 8, 16]) for _ in range(L)]
    dq = rng.choice([16, 24, 40])
    dtype = rng.choice([torch.float16, torch.bfloat16, torch.float32])
    nice = rng.random() < 0.5
    N, Lq = rng.choice([1, 2, 3]), rng.choice([1, 5, 12, 33])
    g = torch.Generator().manual_seed(rng.randrange(1 << 30))
    with contextlib.redirect_stdout(io.StringIO()):
        m = MMFS(d_model=H * D, d_query=dq, d_value=16, d_out=dq, n_levels=L, n_heads=H, n_points=P, ratio=1.0,
                 offset_init_magnitude=rng.choice([1, 3]), spatial_shapes=sizes, base_spatial_shape=rng.choice([2, 4, 8]),
                 max_num_image_per_seq=6)

    def rnd(shape, scale):
        t = torch.randn(shape, generator=g) * scale
        return (t * 64).round() / 64 if nice else t
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(rnd(p.shape, 0.3))
        if not nice:
            m.sampling_offsets.bias.mul_(3.0)
    m = m.to(DEV, dtype).eval()
    shapes, start, S = make_level_tables([(s, s) for s in sizes], n, DEV)
    query = rnd((N, Lq, dq), 1.0).to(DEV, dtype)
    feat = rnd((N, n, S // n, 16), 1.0).to(DEV, dtype)
    ref = (torch.rand(1, Lq, 1, 2, generator=g) * 16).round().div(16).to(DEV, dtype) if nice else torch.rand(1, Lq, 1, 2, generator=g).to(DEV, dtype)
    mask = (torch.rand(N, n, generator=g) < 0.8).float().to(DEV)
    mask[:, 0] = 1
    desc = f"#{idx} {str(dtype)[6:]} H{H} D{D} L{L} P{P} n{n} {sizes} dq{dq} N{N} Lq{Lq} nice={nice}"
    # round 4: at most 8 queries per (sample, head) take mmfs_sample_decode (same products, the fp32 sums in another
    # order: equal within the storage type's rounding); with MMFS_SAMPLE_DECODE=0 they stay on the in-order kernel
    from mmfs_amd.functions.mmfs_plan_func import sample_forward_groups
    os.environ["MMFS_SAMPLE_DECODE"] = rng.choice(["0", "1"])
    import MultiScaleDeformableAttention as _msda
    _msda.reload_env()                       # (the library reads its knobs once: csrc/msda_env.h)
    in_order = sample_forward_groups(dtype, Lq, D, n * L, P) <= 1
    desc += f" decode_kernel={not in_order}"
    outs = {}
    for fused in (True, False):
        m.fused_sampler = fused
        with torch.no_grad():
            outs[fused] = m(query, ref, feat, shapes, start, None, mask)
    if in_order:
        ok = torch.equal(outs[True], outs[False])
    else:
        tol = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
        a, b = outs[True].double(), outs[False].double()
        ok = bool(((a - b).abs().max() <= tol * b.abs().max().clamp_min(1.0)).item())
    ok = ok and bool(torch.isfinite(outs[True]).all())
    if not ok:
        d = (outs[True].float() - outs[False].float()).abs()
        desc += f"  max diff {float(d.max()):.3e} in {int((d > 0).sum())} of {d.numel()}"
    return ok, desc


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    bad = 0
    for i in range(n_cases):
        ok, desc = one_case(rng, i)
        if not ok:
            bad += 1
            print("FAIL", desc, flush=True)
        elif i < 10:
            print("ok  ", desc, flush=True)
    print(f"{n_cases - bad}/{n_cases} cases bit-identical (in-order kernel) / within the storage type's rounding (decode kernel)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
