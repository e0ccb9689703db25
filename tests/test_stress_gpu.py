"""Repeated-run stress of every matrix-core kernel of the op at BASELINE's full sizes.

Round 4's sliced forward (csrc/msda_fwd_q8.hip) produced wrong rows in 0.2-3 % of runs, only with many busy waves per
CU and never under an in-kernel check (profiles/r04_experiments.md r04f-m); what caught it was a scan that ran the
kernel again and again and compared whole tensors (tools/debug/q8_scan.py).  That scan is a test now, for every
formulation that multiplies on the matrix cores:

  forward   "lds"     csrc/msda_fwd_mma.hip   (one product per query; round 3)
            "slices"  csrc/msda_fwd_q8.hip    (8-query tiles, block-diagonal weights; round 4)
            "waves"   csrc/msda_fwd_wq.hip    (a wave per query, weights on a diagonal; round 5, the default at the north star)
  backward  grad_loc / grad_attn "lds"  csrc/msda_taps_mma.hip;  grad_value: csrc/msda_bwd_tile.hip (always on for 16-bit)

Each is launched RUNS times back to back on the SAME inputs at the full size of a BASELINE configuration; every run's
whole output tensor must equal the first run's bit for bit (the kernels have no atomics and no run-to-run freedom in
their summation order), and the first run's whole tensor must agree with the row-gather formulation (csrc/msda_fwd.hip /
msda_bwd.hip -- which the fuzz and golden tests hold to the oracle) within the storage type's bar."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]

pytestmark = pytest.mark.gpu
RUNS = 50
BAR = {torch.float16: 1e-3, torch.bfloat16: 8e-3}

# BASELINE.json configs at full size: (B, Nq, H, D, P, levels per image, images)
NORTH_STAR = (8, 4096, 8, 128, 4, [(64, 64), (32, 32), (16, 16), (8, 8)], 1)          # config 2, the headline shape
SD_BLOCK = (8, 4096, 16, 64, 8, [(64, 64), (32, 32), (16, 16), (8, 8)], 1)            # config 2 / 4, the reference's real head width
LLM_N1 = (4, 2048, 16, 64, 8, [(32, 32), (16, 16), (8, 8)], 1)                        # config 3
LLM_N4 = (4, 2048, 16, 64, 8, [(32, 32), (16, 16), (8, 8)], 4)                        # config 5


def _inputs(cfg, dtype, seed=0):
    B, Nq, H, D, P, shapes, n = cfg
    g = torch.Generator(device="cuda").manual_seed(seed)
    sh = torch.tensor(shapes * n, dtype=torch.long, device="cuda")
    st = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    S, L = int(sh.prod(1).sum()), sh.shape[0]
    value = torch.rand(B, S, H, D, device="cuda", generator=g).to(dtype)
    loc = (torch.rand(B, Nq, H, L, P, 2, device="cuda", generator=g) * 1.1 - 0.05).to(dtype)     # (borders included)
    attn = torch.rand(B, Nq, H, L, P, device="cuda", generator=g) + 1e-5
    attn = (attn / attn.sum((-1, -2), keepdim=True)).to(dtype)
    grad = torch.randn(B, Nq, H * D, device="cuda", generator=g).to(dtype)
    return value, sh, st, loc, attn, grad


def _rel(a, b):
    return float((a.double() - b.double()).abs().max()) / max(1.0, float(b.double().abs().max()))


FWD_CASES = [("lds", NORTH_STAR), ("waves", NORTH_STAR), ("slices", NORTH_STAR), ("slices", LLM_N1), ("slices", SD_BLOCK),
             ("lds", SD_BLOCK), ("slices", LLM_N4)]


@pytest.mark.parametrize("algo,cfg", FWD_CASES, ids=[f"{a}-B{c[0]}Nq{c[1]}H{c[2]}D{c[3]}P{c[4]}n{c[6]}" for a, c in FWD_CASES])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_forward_formulations_repeat_bit_for_bit_at_full_size(algo, cfg, dtype):
    import MultiScaleDeformableAttention as MSDA
    value, sh, st, loc, attn, _ = _inputs(cfg, dtype)
    old = MSDA._fwd_algo
    try:
        MSDA._fwd_algo = "gather"
        ref = MSDA.ms_deform_attn_forward(value, sh, st, loc, attn, 1)
        MSDA._fwd_algo = algo
        outs = [MSDA.ms_deform_attn_forward(value, sh, st, loc, attn, 1) for _ in range(RUNS)]     # back to back, no sync
        torch.cuda.synchronize()
    finally:
        MSDA._fwd_algo = old
    first = outs[0]
    assert bool(torch.isfinite(first).all())
    different = [i for i, o in enumerate(outs) if not torch.equal(o, first)]
    assert not different, f"{algo}: runs {different[:8]} of {RUNS} differ from the first"
    assert _rel(first, ref) <= 0.5 * BAR[dtype], f"{algo}: whole tensor vs the row gather {_rel(first, ref):.3e}"


# ("sorted": round 6, csrc/msda_bwd_taps_sorted.hip -- grad_loc / grad_attn from the grad_value sort's records; the route needs
# a level table the host has verified: the case registers its tables.  A sample's sums do not depend on where the sort put its
# record: bit-equal run to run although the record order is not)
BWD_CASES = [("lds", NORTH_STAR), ("auto", NORTH_STAR), ("auto", SD_BLOCK), ("auto", LLM_N1), ("sorted", NORTH_STAR), ("sorted", LLM_N4)]


@pytest.mark.parametrize("algo,cfg", BWD_CASES, ids=[f"{a}-B{c[0]}Nq{c[1]}H{c[2]}D{c[3]}P{c[4]}n{c[6]}" for a, c in BWD_CASES])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_backward_formulations_repeat_bit_for_bit_at_full_size(algo, cfg, dtype):
    """grad_loc / grad_attn through the LDS-resident kernel (forced, and as the default routing has it) and grad_value through
    the cell sort + matrix-core tile reduce: RUNS backward calls, every gradient tensor bit-equal run to run; the first
    against the row-gather taps kernel and the float-atomic scatter (fp32 sums in another order)."""
    import MultiScaleDeformableAttention as MSDA
    value, sh, st, loc, attn, grad = _inputs(cfg, dtype, seed=1)
    if algo == "sorted":
        assert MSDA.register_level_tables(sh, st, value.shape[1])[0]
    runs = RUNS // 2
    old, old_b = MSDA._taps_algo, MSDA._bwd_algo
    try:
        MSDA._taps_algo, MSDA._bwd_algo = "gather", "atomic"    # row-gather taps + the reference's float-atomic scatter
        ref = MSDA.ms_deform_attn_backward(value, sh, st, loc, attn, grad, 1)
        MSDA._taps_algo, MSDA._bwd_algo = algo, "auto"
        outs = [MSDA.ms_deform_attn_backward(value, sh, st, loc, attn, grad, 1) for _ in range(runs)]
        torch.cuda.synchronize()
    finally:
        MSDA._taps_algo, MSDA._bwd_algo = old, old_b
    first = outs[0]
    for name, i in (("grad_loc", 1), ("grad_attn", 2)):
        assert bool(torch.isfinite(first[i]).all()), name
        different = [r for r, o in enumerate(outs) if not torch.equal(o[i], first[i])]
        assert not different, f"{name} ({algo}): runs {different[:8]} of {runs} differ from the first"
    # grad_value: the cell sort places a cell's records in the order its LDS atomics hand out slots, so the fp32 sums of a
    # pixel run in another order from run to run (the reference's float atomics have the same freedom): equal to ONE rounding
    # step of the storage type (an ulp of the largest elements: <= 2^-7 of them in bf16, 2^-10 in fp16)
    assert bool(torch.isfinite(first[0]).all())
    worst = max(_rel(o[0], first[0]) for o in outs[1:])
    assert worst <= BAR[dtype], f"grad_value ({algo}): {worst:.3e} from run to run"
    # grad_loc is discontinuous where a pixel coordinate crosses an integer (the two formulations agree on which side: the
    # same fp32 expression); the dots are fp32 sums in another order
    assert _rel(first[2], ref[2]) <= BAR[dtype] and _rel(first[1], ref[1]) <= BAR[dtype]
    assert _rel(first[0], ref[0]) <= BAR[dtype]
