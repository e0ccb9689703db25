"""CPU tests that pin the oracle (oracle/) to the reference through the committed
golden vectors (tests/golden/, produced by tests/golden/make_golden.py from the
reference's ms_deform_attn_core_pytorch, ops/functions/ms_deform_attn_func.py:47-67),
plus the kernel-only edge semantics the PyTorch statement does not have
(ms_deform_im2col_cuda.cuh:291)."""
import glob
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, level_tables, load_golden, make_inputs, max_abs
from oracle import msda_oracle, msda_torch

OP_GOLDENS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "op_*.npz")))


def test_goldens_present():
    assert len(OP_GOLDENS) >= 8


@pytest.mark.parametrize("name", OP_GOLDENS)
def test_c_oracle_matches_reference_goldens_fp64(name):
    z = load_golden(name)
    args = (z["value"], z["spatial_shapes"], z["level_start_index"], z["loc"], z["attn"])
    out = msda_oracle.forward(*args)
    gv, gl, ga = msda_oracle.backward(*args, z["grad_out"])
    stored_f32 = z["grad_value_f64"].dtype == np.float32
    assert max_abs(out, z["out_f64"]) < 1e-13
    assert max_abs(gv, z["grad_value_f64"]) < (2e-7 if stored_f32 else 1e-13)
    assert max_abs(gl, z["grad_loc_f64"]) < 1e-12
    assert max_abs(ga, z["grad_attn_f64"]) < 1e-12


@pytest.mark.parametrize("name", [n for n in OP_GOLDENS if n != "op_g3_border"])
def test_c_oracle_fp32_within_1e5_of_reference(name):
    # the fp32 parity bar of BASELINE.json: 1e-5 against the fp64 answer
    z = load_golden(name)
    args = (z["value"], z["spatial_shapes"], z["level_start_index"], z["loc"], z["attn"])
    out = msda_oracle.forward(*args, dtype=np.float32)
    gv, gl, ga = msda_oracle.backward(*args, z["grad_out"], dtype=np.float32)
    assert max_abs(out, z["out_f64"]) < 1e-5
    assert max_abs(gv, z["grad_value_f64"]) < 1e-5
    assert max_abs(ga, z["grad_attn_f64"]) < 2e-5
    # grad_loc scales with the level extent (W*...): relative bar
    assert max_abs(gl, z["grad_loc_f64"]) < 1e-5 * max(1.0, float(np.abs(z["grad_loc_f64"]).max()))


@pytest.mark.parametrize("seed", [0, 1])
def test_c_oracle_matches_torch_restatement(seed):
    x = make_inputs(B=2, H=3, D=8, Nq=7, P=3, shapes=[(5, 4), (3, 6), (1, 2)], seed=seed,
                    loc_range=(-0.2, 1.2), dtype=torch.float64)
    out = msda_oracle.forward(x["value"], x["shapes"], x["start"], x["loc"], x["attn"])
    gv, gl, ga = msda_oracle.backward(x["value"], x["shapes"], x["start"], x["loc"], x["attn"], x["grad"])
    o2, gv2, gl2, ga2 = msda_torch.fwd_bwd(x["value"], x["shapes"].tolist(), x["loc"], x["attn"], x["grad"])
    assert max_abs(out, o2.numpy()) < 1e-13
    assert max_abs(gv, gv2.numpy()) < 1e-13
    assert max_abs(gl, gl2.numpy()) < 1e-12
    assert max_abs(ga, ga2.numpy()) < 1e-12


def test_oracle_gradients_match_finite_differences():
    x = make_inputs(B=1, H=2, D=3, Nq=2, P=2, shapes=[(4, 3), (2, 2)], seed=3, dtype=torch.float64)
    args = [x["value"].numpy(), x["shapes"].numpy(), x["start"].numpy(), x["loc"].numpy(), x["attn"].numpy()]
    g = x["grad"].numpy()
    gv, gl, ga = msda_oracle.backward(*args, g)
    f = lambda: float((msda_oracle.forward(*args) * g.reshape(1, 2, 6)).sum())
    eps = 1e-6
    rng = np.random.default_rng(0)
    for arr, grad in ((args[0], gv), (args[3], gl), (args[4], ga)):
        for _ in range(6):
            idx = tuple(rng.integers(0, s) for s in arr.shape)
            old = arr[idx]
            arr[idx] = old + eps; hi = f()
            arr[idx] = old - eps; lo = f()
            arr[idx] = old
            assert abs((hi - lo) / (2 * eps) - grad[idx]) < 1e-6


def test_kernel_edge_semantics():
    """Strict range test, NaN/Inf locations and out-of-map corners (cuh:58-81, 291)."""
    sh, start = level_tables([(2, 2)])
    value = torch.arange(1.0, 5.0, dtype=torch.float64).reshape(1, 4, 1, 1)   # [[1,2],[3,4]]
    def run(x, y):
        loc = torch.tensor([x, y], dtype=torch.float64).reshape(1, 1, 1, 1, 1, 2)
        attn = torch.ones(1, 1, 1, 1, 1, dtype=torch.float64)
        out = msda_oracle.forward(value, sh, start, loc, attn)
        g = msda_oracle.backward(value, sh, start, loc, attn, np.ones((1, 1, 1)))
        return float(out.reshape(())), g
    assert run(0.25, 0.25)[0] == 1.0                  # pixel centre (0,0)
    assert run(0.75, 0.75)[0] == 4.0
    assert run(0.5, 0.5)[0] == 2.5                    # mean of the four
    assert run(0.0, 0.0)[0] == 0.25                   # corner: only (0,0) in range, weight 1/4
    assert run(1.0, 1.0)[0] == 1.0                    # 4 * 0.25
    for bad in (float("nan"), float("inf"), -float("inf")):
        out, (gv, gl, ga) = run(bad, 0.5)
        assert out == 0.0 and not gv.any() and not gl.any() and not ga.any()
    # x_im = loc*W - 0.5 == -1 exactly at loc = -0.25 (W = 2): fails the strict test
    out, (gv, gl, ga) = run(-0.25, 0.5)
    assert out == 0.0 and not gv.any() and not gl.any() and not ga.any()
    # x_im == W at loc = 1.25: fails; just inside still contributes nothing (both right corners out)
    assert run(1.25, 0.5)[0] == 0.0
    out, (gv, gl, ga) = run(1.2499, 0.5)
    assert 0.0 < out < 1e-3 and gl.reshape(-1)[0] < 0    # moving right loses the last column


def test_empty_and_ragged_shapes():
    sh, start = level_tables([(3, 1), (1, 5)])
    v = np.random.default_rng(0).random((2, 8, 2, 4))
    loc = np.zeros((2, 0, 2, 2, 3, 2)); attn = np.zeros((2, 0, 2, 2, 3))
    assert msda_oracle.forward(v, sh, start, loc, attn).shape == (2, 0, 8)
    loc = np.random.default_rng(1).random((2, 3, 2, 2, 3, 2)); attn = np.random.default_rng(2).random((2, 3, 2, 2, 3))
    out = msda_oracle.forward(v, sh, start, loc, attn)
    o2 = msda_torch.msda_grid_sample(torch.from_numpy(v), sh.tolist(), torch.from_numpy(loc), torch.from_numpy(attn))
    assert max_abs(out, o2.numpy()) < 1e-13
