"""GPU parity of the backward on the cell-sorted records (mmfs_msda_backward_sorted, csrc/msda_bwd_taps_sorted.hip):
grad_loc / grad_attn from the records the grad_value sort makes, against the CPU oracle (reference:
ms_deform_im2col_cuda.cuh:90-162).  The route needs a level table the host has verified, so every case registers its
tables; ``_taps_algo = "sorted"`` makes an argument set the route does not take an error instead of a silent fallback."""
import numpy as np
import pytest
import torch

from helpers import make_inputs, max_abs
from test_op_gpu import check, run_hip, run_oracle

pytestmark = pytest.mark.gpu

CASES = [
    # B, H, D, Nq, P, shapes                                          what it exercises
    (2, 8, 128, 257, 4, [(16, 16), (8, 8), (4, 4), (2, 2)]),           # north-star head width, extents that are multiples of 4
    (1, 8, 128, 300, 4, [(64, 64), (32, 32), (16, 16), (8, 8)]),       # the north-star pyramid
    (2, 4, 64, 333, 8, [(32, 32), (16, 16), (8, 8)] * 2),              # LLM geometry, two images, P = 8 (two vectors per query)
    (1, 16, 64, 200, 8, [(64, 64), (32, 32), (16, 16), (8, 8)]),       # SD block geometry
    (1, 4, 32, 513, 4, [(16, 16), (3, 3)]),                            # D = 32 (one product per chain), a level inside ONE block
    (1, 2, 64, 97, 16, [(1, 1), (2, 9), (16, 16)]),                    # P = 16 (grouped scan), degenerate levels
    (2, 3, 128, 150, 4, [(5, 7), (6, 5), (13, 9), (3, 2)]),            # extents not multiples of 4: ragged last blocks
    (1, 8, 32, 4096, 4, [(16, 13), (40, 6)]),                          # as many queries as one trip of the sort takes
    (1, 2, 128, 60, 4, [(9, 9)] * 20),                                 # L = 20
]
IDS = [f"B{c[0]}H{c[1]}D{c[2]}Nq{c[3]}P{c[4]}L{len(c[5])}" for c in CASES]


@pytest.fixture
def sorted_route(monkeypatch):
    import MultiScaleDeformableAttention as MSDA
    monkeypatch.setattr(MSDA, "_taps_algo", "sorted")
    log = []
    monkeypatch.setattr(MSDA, "_event_log", log)
    yield log
    monkeypatch.setattr(MSDA, "_event_log", None)


def ran_sorted(log):
    names = [n for n, _, _ in log]
    assert names[-4:] == ["msda_bwd_value_prepare", "msda_bwd_value_sort", "msda_bwd_taps", "msda_bwd_value_reduce"], names


@pytest.mark.parametrize("case", CASES, ids=IDS)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_sorted_backward_matches_oracle(case, dtype, sorted_route):
    B, H, D, Nq, P, shapes = case
    x = make_inputs(B, H, D, Nq, P, shapes, seed=21, loc_range=(-0.15, 1.15), dtype=dtype)
    x["loc"][0, 3, 0, 0, 0, 0] = float("nan")          # non-finite locations contribute nothing and get zero gradients
    x["loc"][0, 5, 1 % H, -1, 0, 1] = float("inf")
    got = run_hip(x, dtype, use_autograd=False, register=True)
    ran_sorted(sorted_route)
    check(got, run_oracle(x), dtype, f"sorted {case[:5]}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_sorted_backward_exact_pixel_centres_and_borders(dtype, sorted_route):
    """Locations on pixel centres (zero fractions), on the map's edges and just outside: every selection branch of the
    5x5 -- first / last row and column of a block, the border cells of the last block -- with grad_attn of zero weights."""
    B, H, D, Nq, P, shapes = 1, 2, 64, 128, 4, [(8, 8), (6, 10)]
    x = make_inputs(B, H, D, Nq, P, shapes, seed=5, dtype=dtype)
    g = torch.Generator().manual_seed(3)
    for l, (Hl, Wl) in enumerate(shapes):
        ys = (torch.randint(-1, Hl + 1, (B, Nq, H, P), generator=g).double() + 0.5) / Hl
        xs = (torch.randint(-1, Wl + 1, (B, Nq, H, P), generator=g).double() + 0.5) / Wl
        half = torch.rand(B, Nq, H, P, generator=g) < 0.5
        x["loc"][:, :, :, l, :, 0] = torch.where(half, xs, x["loc"][:, :, :, l, :, 0])
        x["loc"][:, :, :, l, :, 1] = torch.where(half, ys, x["loc"][:, :, :, l, :, 1])
    x["loc"] = x["loc"].to(dtype).double()
    x["attn"][:, ::3] = 0.0                              # zero weights: grad_attn is not zero, grad_value gets nothing
    got = run_hip(x, dtype, use_autograd=False, register=True)
    ran_sorted(sorted_route)
    want = run_oracle(x)
    # grad_loc is discontinuous where a pixel coordinate is an integer (the fp32 product lands on either side): compare
    # it away from the crossings only, everything else everywhere
    loc = x["loc"].numpy()
    sh = np.asarray(shapes, dtype=np.float64)
    px = loc[..., 0] * sh[None, None, None, :, None, 1] - 0.5
    py = loc[..., 1] * sh[None, None, None, :, None, 0] - 0.5
    near = (np.abs(px - np.round(px)) < 1e-3) | (np.abs(py - np.round(py)) < 1e-3)
    gl_got, gl_want = got[2].copy(), np.asarray(want[2], dtype=np.float64).reshape(got[2].shape).copy()
    gl_got[near] = 0.0
    gl_want[near] = 0.0
    check([got[0], got[1], gl_got, got[3]], [want[0], want[1], gl_want, want[3]], dtype, "centres")


@pytest.mark.parametrize("dtype", [torch.bfloat16])
def test_sorted_backward_lazy_zero_attn(dtype, sorted_route):
    """MMFS's hint (zero weights never have their gradients read): no record, zeros written by the sort."""
    import MultiScaleDeformableAttention as MSDA
    B, H, D, Nq, P, shapes = 2, 4, 128, 100, 4, [(16, 16), (8, 8), (4, 4)]
    x = make_inputs(B, H, D, Nq, P, shapes, seed=9, loc_range=(-0.1, 1.1), dtype=dtype)
    x["attn"][:, :, :, 1] = 0.0
    dev = lambda t: t.to("cuda", dtype) if t.is_floating_point() else t.to("cuda")
    value, loc, attn, grad = dev(x["value"]), dev(x["loc"]), dev(x["attn"]), dev(x["grad"])
    sh, st = dev(x["shapes"]), dev(x["start"])
    MSDA.register_level_tables(sh, st, value.shape[1], host_shapes=x["shapes"], host_start=x["start"])
    gv, gl, ga = MSDA.ms_deform_attn_backward(value, sh, st, loc, attn, grad.reshape(B, Nq, H * D), 1, lazy_zero_attn=True)
    torch.cuda.synchronize()
    ran_sorted(sorted_route)
    want = run_oracle(x)
    zero = (x["attn"] == 0).numpy()
    w_gl = np.asarray(want[2], dtype=np.float64).reshape(gl.shape).copy()
    w_ga = np.asarray(want[3], dtype=np.float64).reshape(ga.shape).copy()
    w_gl[zero] = 0.0
    w_ga[zero] = 0.0
    got = [np.asarray(want[0]), gv.double().cpu().numpy(), gl.double().cpu().numpy(), ga.double().cpu().numpy()]
    assert np.all(got[3][zero] == 0.0) and np.all(got[2][zero] == 0.0)
    check(got, [want[0], want[1], w_gl, w_ga], dtype, "lazy")


def test_sorted_backward_non_finite_value_rows_stay_with_their_samples(sorted_route):
    """A non-finite value row reaches the gradients of the samples whose footprint holds it and no other (the selection
    is by conditional move, never a multiplication by zero): element for element the oracle's Inf / NaN pattern."""
    dtype = torch.bfloat16
    B, H, D, Nq, P, shapes = 1, 2, 128, 64, 4, [(8, 8), (4, 4)]
    x = make_inputs(B, H, D, Nq, P, shapes, seed=13, dtype=dtype)
    x["value"][0, 3 * 8 + 4, 0, 5] = float("inf")      # pixel (3, 4) of level 0, head 0
    x["value"][0, 64 + 5, 1, 0] = float("nan")         # pixel (1, 1) of level 1, head 1
    got = run_hip(x, dtype, use_autograd=False, register=True)
    ran_sorted(sorted_route)
    want = run_oracle(x)
    for k in (2, 3):
        w = np.asarray(want[k], dtype=np.float64).reshape(got[k].shape)
        assert np.array_equal(np.isfinite(got[k]), np.isfinite(w)), ("grad_loc", "grad_attn")[k - 2]
        fin = np.isfinite(w)
        scale = max(1.0, float(np.abs(w[fin]).max()))
        assert max_abs(got[k][fin], w[fin]) <= 8e-3 * scale
