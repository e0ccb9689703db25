"""GPU: the one-pass feature-bank kernel (csrc/mmfs_bank.hip) against the framework-op
statement of the same steps (gather over images, "c (h w) -> (h w) c", concatenation over levels, zero
slots -- mm_interleaved.py:223-250).  A copy: the bar is bit-equality; the backward sums in fp32."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def torch_bank(levels, src):
    packed = torch.cat([f.flatten(2).transpose(1, 2) for f in levels], dim=1)
    valid = (src >= 0) & (src < packed.shape[0])
    rows = packed.index_select(0, src.clamp(0, packed.shape[0] - 1))
    return rows * valid[:, None, None].to(rows.dtype)


CASES = [
    # n_img, C, level shapes, slot indices
    (5, 64, [(32, 32), (16, 16), (8, 8)], [0, 1, -1, 4, 2, 3, -1, -1]),            # LLM bank, padded slots
    (3, 1024, [(64, 64), (32, 32), (16, 16), (8, 8)], [2, 0, 1]),                  # SD pack, real widths
    (4, 24, [(14, 14), (7, 7)], [3, 3, 0, 7, -5, 1]),                              # odd sizes: scalar paths; repeats; out of range
    (2, 72, [(5, 9), (1, 1), (3, 70)], [1, 0, 1]),                                 # ragged tiles in both directions
    (1, 8, [(1, 1)], [0]),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("case", CASES, ids=[f"img{c[0]}C{c[1]}L{len(c[2])}" for c in CASES])
def test_bank_gather_is_a_bit_exact_copy(case, dtype):
    from mmfs_amd.functions import BankGatherFunction, bank_gather_supported
    n_img, C, shapes, slots = case
    g = torch.Generator().manual_seed(7)
    levels = [torch.randn(n_img, C, h, w, generator=g).to(DEV, dtype) for h, w in shapes]
    src = torch.tensor(slots, device=DEV)
    assert bank_gather_supported(levels, src.numel())
    got = BankGatherFunction.apply(src, *levels)
    want = torch_bank(levels, src)
    assert got.shape == want.shape and torch.equal(got, want)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("case", CASES[:4], ids=[f"img{c[0]}C{c[1]}L{len(c[2])}" for c in CASES[:4]])
def test_bank_scatter_is_the_adjoint(case, dtype):
    from mmfs_amd.functions import BankGatherFunction
    n_img, C, shapes, slots = case
    g = torch.Generator().manual_seed(9)
    base = [torch.randn(n_img, C, h, w, generator=g).to(DEV, dtype) for h, w in shapes]
    src = torch.tensor(slots, device=DEV)
    lv_a = [f.clone().requires_grad_(True) for f in base]
    lv_b = [f.clone().double().requires_grad_(True) for f in base]
    out = BankGatherFunction.apply(src, *lv_a)
    grad = torch.randn(out.shape, generator=g).to(DEV, dtype)
    out.backward(grad)
    torch_bank(lv_b, src).backward(grad.double())
    repeats = len(set(s for s in slots if 0 <= s < n_img)) < sum(0 <= s < n_img for s in slots)
    for a, b in zip(lv_a, lv_b):
        want = b.grad.to(dtype)                     # fp64 sum rounded once == fp32 sum rounded once for <= a few terms
        if not repeats:
            assert torch.equal(a.grad, want)
        else:
            tol = 1e-6 if dtype == torch.float32 else 8e-3
            assert float((a.grad.double() - b.grad).abs().max()) <= tol * max(1.0, float(b.grad.abs().max()))
        shown = torch.zeros(n_img, dtype=torch.bool)
        for s in slots:
            if 0 <= s < n_img:
                shown[s] = True
        assert float(a.grad[~shown.to(DEV)].abs().sum()) == 0.0      # images nobody shows: exact zeros


def test_builders_use_the_kernel_and_match_the_framework_ops():
    """prepare_mmfs_features_for_mm_decoder / pack_image_levels / MMFSNet._pack on device tensors go
    through BankGatherFunction and give the bank the host statement gives."""
    import MultiScaleDeformableAttention as MSDA
    from mmfs_amd import bank
    from mmfs_amd.blocks import MMFSNet
    g = torch.Generator().manual_seed(3)
    num = torch.tensor([2, 0, 3])
    levels = [torch.randn(5, 32, s, s, generator=g).to(torch.bfloat16) for s in (16, 8, 4)]
    want = bank.llm_feature_bank_from_levels(levels, num, 3)                       # host tensors: framework ops
    MSDA._event_log = log = []
    try:
        got = bank.llm_feature_bank_from_levels([f.to(DEV) for f in levels], num.to(DEV), 3)
        packed = bank.pack_image_levels([f.to(DEV) for f in levels])
        net_bank = MMFSNet._pack([f.to(DEV)[:4].reshape(2, 2, 32, f.shape[-1], f.shape[-1]) for f in levels])
    finally:
        MSDA._event_log = None
    assert [n for n, _, _ in log] == ["mmfs_bank_gather"] * 3
    assert torch.equal(got.cpu(), want)
    assert torch.equal(packed.cpu(), bank.pack_image_levels(levels))
    assert torch.equal(net_bank.cpu(), MMFSNet._pack([f[:4].reshape(2, 2, 32, f.shape[-1], f.shape[-1]) for f in levels]))
