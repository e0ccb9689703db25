"""Fused MMFS sampling plan: one gfx950 kernel each way for everything the
reference's MMFS.forward does between its Linear layers and the op
(mm_interleaved/models/utils/ops/modules/mmfs.py:181-265).  C ABI: ``mmfs_plan_forward`` /
``mmfs_plan_backward`` in include/mmfs_msda.h; kernels in csrc/mmfs_plan.hip.

``mmfs_plan_supported`` tells the module whether the fused path applies (device tensors, 2-D
reference points shared by all levels, P in {4, 8, 16}, n_images * n_levels <= 64); otherwise the module evaluates the same
mathematics with framework ops -- that is NOT a CPU fallback of the sampling op, only of this
front-end, and it is what the CPU parity tests of the module exercise.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import MultiScaleDeformableAttention as MSDA

_lib = MSDA._lib
_i64, _vp, _int = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
_lib.mmfs_plan_forward.restype = _int
_lib.mmfs_plan_forward.argtypes = [_int] + [_vp] * 11 + [_i64] * 9 + [_vp]
_lib.mmfs_plan_backward.restype = _int
_lib.mmfs_plan_backward.argtypes = [_int] + [_vp] * 12 + [_i64] * 9 + [_vp]
_lib.mmfs_sample_forward.restype = _int
_lib.mmfs_sample_forward.argtypes = [_int] + [_vp] * 12 + [_i64] * 11 + [_vp]
_lib.mmfs_sample_forward_token.restype = _int
_lib.mmfs_sample_forward_token.argtypes = [_int] + [_vp] * 13 + [_i64] * 11 + [_vp]
_lib.mmfs_sample_forward_heads.restype = _int
_lib.mmfs_sample_forward_heads.argtypes = [_int] + [_vp] * 5 + [_i64] * 2 + [_vp] * 8 + [_i64] * 11 + [_vp]
_lib.mmfs_plan_forward_heads.restype = _int
_lib.mmfs_plan_forward_heads.argtypes = [_int, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64] + [_vp] * 7 + [_i64] * 9 + [_vp]
_lib.mmfs_plan_backward_heads.restype = _int
_lib.mmfs_plan_backward_heads.argtypes = [_int] + [_vp] * 8 + [_vp, _vp, _i64, _i64, _int, _vp, _vp, _i64, _i64] + [_i64] * 9 + [_vp]
_lib.mmfs_sample_forward_groups.restype = _int
_lib.mmfs_sample_forward_groups.argtypes = [_int] + [_i64] * 4
_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def mmfs_plan_supported(query, reference_points, n_levels, n_points, n_images):
    return (query.is_cuda and query.dtype in _CODE and reference_points.shape[-1] == 2
            and reference_points.shape[2] == 1 and n_points in (4, 8, 16) and n_images * n_levels <= 64)


class MMFSPlanFunction(Function):
    """(off_q [N,Lq,H*P*2], att_q [N,Lq,H*L*P], off_tab [M,H*P*2], att_tab [M,H*L*P]  -- point columns
    only, the sink column of the attention head is a constant and is never evaluated --
    relpos [N,Lr,n] long, ref [Nr,Lq,2] fp32, shapes [n*L,2] long, ratios [L] fp32, H, L, P)
    -> loc [N,Lq,H,n*L,P,2], attn [N,Lq,H,n*L,P], sink [N,Lq,H] (fp32)."""

    @staticmethod
    def forward(ctx, off_q, att_q, off_tab, att_tab, relpos, ref, shapes, ratios, H, L, P):
        dt = off_q.dtype
        N, Lq = off_q.shape[0], off_q.shape[1]
        n, Lr, Nr, M = relpos.shape[-1], relpos.shape[1], ref.shape[0], off_tab.shape[0]
        off_q, att_q = off_q.contiguous(), att_q.to(dt).contiguous()
        off_tab, att_tab = off_tab.to(dt).contiguous(), att_tab.to(dt).contiguous()
        relpos, ref, ratios = relpos.contiguous(), ref.float().contiguous(), ratios.float().contiguous()
        assert shapes.dtype == torch.int64 and shapes.is_contiguous() and shapes.shape == (n * L, 2), "spatial_shapes must be a contiguous int64 [n*L, 2] tensor"
        dev = off_q.device
        loc = torch.empty((N, Lq, H, n * L, P, 2), dtype=dt, device=dev)
        attn = torch.empty((N, Lq, H, n * L, P), dtype=dt, device=dev)
        sink = torch.empty((N, Lq, H), dtype=torch.float32, device=dev)
        dims = (N, Lq, H, L, P, n, M, Lr, Nr)
        with torch.cuda.device(dev):
            rc = MSDA._launch("mmfs_plan_fwd", dev, _lib.mmfs_plan_forward, _CODE[dt], off_q.data_ptr(),
                              att_q.data_ptr(), off_tab.data_ptr(), att_tab.data_ptr(), relpos.data_ptr(),
                              ref.data_ptr(), shapes.data_ptr(), ratios.data_ptr(), loc.data_ptr(),
                              attn.data_ptr(), sink.data_ptr(), *dims, MSDA._stream(dev))
        MSDA._check(rc, "mmfs_plan_forward")
        ctx.save_for_backward(attn, sink, relpos, shapes, ratios)
        ctx.dims = dims
        ctx.shapes_in = (off_q.shape, att_q.shape, off_tab.shape, att_tab.shape)
        return loc, attn, sink

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loc, g_attn, g_sink):
        attn, sink, relpos, shapes, ratios = ctx.saved_tensors
        dt, dev = attn.dtype, attn.device
        N, Lq, H, L, P, n, M, Lr, Nr = ctx.dims
        g_loc = g_loc.to(dt).contiguous()
        g_attn = g_attn.to(dt).contiguous()
        g_sink = g_sink.float().contiguous() if g_sink is not None else None
        d_off_q = torch.empty((N, Lq, H * P * 2), dtype=torch.float32, device=dev)
        d_att_q = torch.empty((N, Lq, H * L * P), dtype=torch.float32, device=dev)
        d_off_tab = torch.zeros((M, H * P * 2), dtype=torch.float32, device=dev)
        d_att_tab = torch.zeros((M, H * L * P), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = MSDA._launch("mmfs_plan_bwd", dev, _lib.mmfs_plan_backward, _CODE[dt], g_loc.data_ptr(),
                              g_attn.data_ptr(), g_sink.data_ptr() if g_sink is not None else None,
                              attn.data_ptr(), sink.data_ptr(), relpos.data_ptr(), shapes.data_ptr(),
                              ratios.data_ptr(), d_off_q.data_ptr(), d_att_q.data_ptr(), d_off_tab.data_ptr(),
                              d_att_tab.data_ptr(), N, Lq, H, L, P, n, M, Lr, Nr, MSDA._stream(dev))
        MSDA._check(rc, "mmfs_plan_backward")
        s0, s1, s2, s3 = ctx.shapes_in
        return (d_off_q.to(dt).reshape(s0), d_att_q.to(dt).reshape(s1), d_off_tab.to(dt).reshape(s2),
                d_att_tab.to(dt).reshape(s3), None, None, None, None, None, None, None)


class MMFSHeadsPlanFunction(Function):
    """``MMFSPlanFunction`` for a caller that evaluates the two query heads as ONE GEMM and the two tables as one
    (``MMFS.sampling_plan`` with gradients): both [N, Lq, H*P*2 + H*L*P] -- the stacked heads' result, offsets' columns
    first --, tabs [M, H*P*2 + H*L*P] likewise; the kernels read the column ranges as they lie (C ABI
    ``mmfs_plan_forward_heads`` / ``mmfs_plan_backward_heads``).  The backward returns ONE gradient per stacked tensor,
    the query-side one written by the kernel in the storage type: against the packed Function's four fp32 tensors, four
    casts and (in autograd) two zero-filled buffers with two slice copies and an add per stacked tensor.
    -> loc [N,Lq,H,n*L,P,2], attn [N,Lq,H,n*L,P], sink [N,Lq,H] (fp32)."""

    @staticmethod
    def forward(ctx, both, tabs, relpos, ref, shapes, ratios, H, L, P):
        dt = both.dtype
        N, Lq, C = both.shape
        n_off = H * P * 2
        assert C == n_off + H * L * P and tabs.shape[1] == C and tabs.dtype == dt
        n, Lr, Nr, M = relpos.shape[-1], relpos.shape[1], ref.shape[0], tabs.shape[0]
        both, tabs = both.contiguous(), tabs.contiguous()
        relpos, ref, ratios = relpos.contiguous(), ref.float().contiguous(), ratios.float().contiguous()
        assert shapes.dtype == torch.int64 and shapes.is_contiguous() and shapes.shape == (n * L, 2), "spatial_shapes must be a contiguous int64 [n*L, 2] tensor"
        dev, es = both.device, both.element_size()
        loc = torch.empty((N, Lq, H, n * L, P, 2), dtype=dt, device=dev)
        attn = torch.empty((N, Lq, H, n * L, P), dtype=dt, device=dev)
        sink = torch.empty((N, Lq, H), dtype=torch.float32, device=dev)
        dims = (N, Lq, H, L, P, n, M, Lr, Nr)
        with torch.cuda.device(dev):
            rc = MSDA._launch("mmfs_plan_fwd", dev, _lib.mmfs_plan_forward_heads, _CODE[dt], both.data_ptr(),
                              both.data_ptr() + n_off * es, C, C, tabs.data_ptr(), tabs.data_ptr() + n_off * es, C, C,
                              relpos.data_ptr(), ref.data_ptr(), shapes.data_ptr(), ratios.data_ptr(), loc.data_ptr(),
                              attn.data_ptr(), sink.data_ptr(), *dims, MSDA._stream(dev))
        MSDA._check(rc, "mmfs_plan_forward_heads")
        ctx.save_for_backward(attn, sink, relpos, shapes, ratios)
        ctx.dims = dims
        return loc, attn, sink

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loc, g_attn, g_sink):
        attn, sink, relpos, shapes, ratios = ctx.saved_tensors
        dt, dev = attn.dtype, attn.device
        N, Lq, H, L, P, n, M, Lr, Nr = ctx.dims
        n_off, C = H * P * 2, H * P * 2 + H * L * P
        g_loc = g_loc.to(dt).contiguous()
        g_attn = g_attn.to(dt).contiguous()
        g_sink = g_sink.float().contiguous() if g_sink is not None else None
        d_both = torch.empty((N, Lq, C), dtype=dt, device=dev)
        d_tabs = torch.zeros((M, C), dtype=torch.float32, device=dev)
        es = d_both.element_size()
        with torch.cuda.device(dev):
            rc = MSDA._launch("mmfs_plan_bwd", dev, _lib.mmfs_plan_backward_heads, _CODE[dt], g_loc.data_ptr(),
                              g_attn.data_ptr(), g_sink.data_ptr() if g_sink is not None else None,
                              attn.data_ptr(), sink.data_ptr(), relpos.data_ptr(), shapes.data_ptr(),
                              ratios.data_ptr(), d_both.data_ptr(), d_both.data_ptr() + n_off * es, C, C, 1,
                              d_tabs.data_ptr(), d_tabs.data_ptr() + n_off * 4, C, C,
                              N, Lq, H, L, P, n, M, Lr, Nr, MSDA._stream(dev))
        MSDA._check(rc, "mmfs_plan_backward_heads")
        return d_both, d_tabs.to(dt), None, None, None, None, None, None, None


def _token_rows(t, vec):
    """[N, Lq, cols] -> (tensor, elements between two token rows; 0 = packed): a column range of a wider row-major
    matrix is handed over as it lies when its rows allow the kernel's vector loads of ``vec`` elements."""
    if t.is_contiguous():
        return t, 0
    if t.dim() == 3 and t.stride(2) == 1:
        ld = t.stride(1) if t.shape[1] > 1 else t.stride(0)
        by = vec * t.element_size()
        if ((t.shape[1] == 1 or t.stride(0) == t.shape[1] * ld) and ld >= t.shape[2] and (ld * t.element_size()) % by == 0
                and t.data_ptr() % by == 0):
            return t, ld
    return t.contiguous(), 0


def sample_forward_groups(dtype, Lq, D, nL, P):
    """Lane groups that share one query's samples in ``mmfs_sample_forward`` for this shape (include/mmfs_msda.h
    ``mmfs_sample_forward_groups``): 1 = summed in sample order, bit-identical to the two kernels; > 1 = a decode-sized
    call (at most 8 queries per (sample, head)), equal within one rounding of the storage type; 0 = not served."""
    return int(_lib.mmfs_sample_forward_groups(_CODE[dtype], Lq, D, nL, P)) if dtype in _CODE else 0


def mmfs_sample_forward(value, shapes, start, off_q, att_q, off_tab, att_tab, relpos, ref, ratios, H, L, P, token=None):
    """Plan -> sampler in ONE kernel (``mmfs_sample_forward`` in include/mmfs_msda.h):
    the locations / weights [N, Lq, H, n*L, P(, 2)] are never written.  Inference only (no autograd graph);
    bit-identical to ``MMFSPlanFunction`` + ``MSDeformAttnFunction`` (decode-sized calls: the same products with the fp32
    sums in another order, see ``sample_forward_groups``).  value [N, S, H, D]; the other
    arguments as for ``MMFSPlanFunction``.  Returns (out [N, Lq, H*D], sink [N, Lq, H] fp32), or None when
    the shape is outside the fused kernel's range (the caller then runs the two kernels).  ``token`` [.., H, D]
    (MMFS's ignore token): ``out + token * sink`` is formed inside the kernel, with the framework statement's roundings."""
    dt = value.dtype
    if dt not in _CODE or off_q.dtype != dt or not value.is_cuda:
        return None
    N, S, Hh, D = value.shape
    Lq = off_q.shape[1]
    n, Lr, Nr, M = relpos.shape[-1], relpos.shape[1], ref.shape[0], off_tab.shape[0]
    value = value.contiguous()
    off_q, ld_off = _token_rows(off_q, 2 * P)
    att_q, ld_att = _token_rows(att_q.to(dt), P)
    off_tab, att_tab = off_tab.to(dt).contiguous(), att_tab.to(dt).contiguous()
    relpos, ref, ratios = relpos.contiguous(), ref.float().contiguous(), ratios.float().contiguous()
    assert shapes.dtype == torch.int64 and shapes.is_contiguous() and shapes.shape == (n * L, 2)
    assert start.dtype == torch.int64 and start.is_contiguous() and start.numel() == n * L
    dev = value.device
    out = torch.empty((N, Lq, Hh * D), dtype=dt, device=dev)
    sink = torch.empty((N, Lq, Hh), dtype=torch.float32, device=dev)
    tok = None
    if token is not None:
        if token.dtype != dt or token.numel() != Hh * D:
            return None
        tok = MSDA._aligned(token.reshape(Hh, D).contiguous())
    with torch.cuda.device(dev):
        rc = MSDA._launch("mmfs_sample_fwd", dev, _lib.mmfs_sample_forward_heads, _CODE[dt], value.data_ptr(),
                          shapes.data_ptr(), start.data_ptr(), off_q.data_ptr(), att_q.data_ptr(), ld_off, ld_att,
                          off_tab.data_ptr(), att_tab.data_ptr(), relpos.data_ptr(), ref.data_ptr(),
                          ratios.data_ptr(), tok.data_ptr() if tok is not None else None, out.data_ptr(), sink.data_ptr(),
                          N, S, Lq, Hh, D, L, P, n, M, Lr, Nr, MSDA._stream(dev))
    if rc == MSDA._E_UNSUPPORTED:
        return None
    MSDA._check(rc, "mmfs_sample_forward")
    return out, sink
