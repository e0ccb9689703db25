"""Autograd nodes for what stands around an MMFS layer's op with gradients enabled (round 4): each replaces a
chain of framework kernels over [tokens, features] tensors by the products that are needed anyway.

``GatedProjectionFunction``  output projection + tanh(gate) + the decoder layer's residual sum (modeling_llama_mmfs.py:346-367,
                             700-717) as one node
``IgnoreTokenFunction``      out + ignore_token * sink per head (mmfs.py:236-241, 274) as one block-diagonal product each way

Plain framework ops inside (GEMMs, one elementwise kernel): they run on any device; the CPU tests check their gradients
against autograd's of the framework statements in fp64.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable


class GatedProjectionFunction(Function):
    """(x [..., K], weight [N, K], bias [N] | None, g [1], residual [..., N]) -> residual + g * (x W^T + bias): an MMFS
    layer's output projection, its tanh(gate) and the decoder layer's residual sum (modeling_llama_mmfs.py:346-367,
    700-717) as one node of the graph.  Forward: the GEMM and ONE elementwise kernel (the framework: a multiply and an
    add).  Backward: NO pass over a [tokens, N] tensor besides the GEMMs' own -- the gate is applied to the SMALL side:
        d x = grad (g W),   d W = g (grad^T x),   d bias = g sum(grad),   d residual = grad,
        d g = sum(grad * (x W^T + bias)) = sum((grad^T x) * W) + sum(sum(grad) * bias)
    where the framework multiplies grad by g (a pass), multiplies grad by the projection's output and reduces it (two
    passes, and the output kept for it).  Same mathematics; the roundings differ by where g meets 16-bit storage."""

    @staticmethod
    def forward(ctx, x, weight, bias, g, residual):
        import torch.nn.functional as F
        y = F.linear(x, weight, bias)
        ctx.save_for_backward(x, weight, bias, g)
        return torch.addcmul(residual, y, g.to(y.dtype))

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        x, weight, bias, g = ctx.saved_tensors
        need = ctx.needs_input_grad
        N, K = weight.shape
        g2, x2 = grad.reshape(-1, N), x.reshape(-1, K)
        gs = g.to(weight.dtype)
        dx = (g2 @ (weight * gs)).reshape(x.shape) if need[0] else None
        dw0 = None
        if need[1] or need[3]:
            from .linear_func import _split_k
            S = _split_k(g2.shape[0], N, K) if (g2.is_cuda and g2.is_contiguous() and x2.is_contiguous()) else 1
            dw0 = (torch.bmm(g2.view(S, -1, N).transpose(1, 2), x2.view(S, -1, K)).sum(0) if S > 1 else g2.t() @ x2)
        db0 = g2.sum(0) if bias is not None and (need[2] or need[3]) else None
        dg = None
        if need[3]:
            acc = torch.promote_types(weight.dtype, torch.float32)
            dg = _dot(dw0.reshape(-1), weight.reshape(-1)).to(acc)
            if db0 is not None:
                dg = dg + _dot(db0, bias).to(acc)
            dg = dg.to(g.dtype).reshape(g.shape)
        dw = dw0 * gs if need[1] else None
        db = db0 * gs if (bias is not None and need[2]) else None
        return dx, dw, db, dg, (grad if need[4] else None)


_dot_ok = {}


def _dot(a, b):
    """sum(a * b) as the BLAS library's dot product where it has one for the type (one pass, fp32 accumulation), else
    a multiply and a reduction."""
    key = (a.dtype, a.device.type)
    ok = _dot_ok.get(key)
    if ok is None:
        try:
            torch.dot(a[:8], b[:8])
            ok = True
        except RuntimeError:
            ok = False
        _dot_ok[key] = ok
    return torch.dot(a, b) if ok else (a * b).sum()


class IgnoreTokenFunction(Function):
    """(out [T, H*D], token [H, D], sink [T, H]) -> out + token * sink, per head (MMFS's ignore token takes the sinks' share of
    the attention: mmfs.py:236-241, 274) as ONE product with a block-diagonal [H, H*D] matrix of the token's rows each way:
        forward   out + sink @ blockdiag(token)          (a GEMM with K = H, reads and writes ``out`` once)
        backward  d sink = grad @ blockdiag(token)^T      (N = H),   d out = grad,   d token = diag blocks of sink^T @ grad
    where the framework statement is a broadcast multiply into a [T, H*D] temporary and an add forward, a multiply and a
    reduction over D backward -- four passes over [T, H*D] tensors for a token the reference initialises to zero and freezes.
    fp32 accumulation, one rounding of the sum (the framework rounds the product first)."""

    @staticmethod
    def forward(ctx, out, token, sink):
        H, D = token.shape
        eye = torch.eye(H, dtype=token.dtype, device=token.device)
        bd = (eye[:, :, None] * token[None]).reshape(H, H * D)
        s2 = sink.to(out.dtype)
        ctx.save_for_backward(bd, s2)
        ctx.hd = (H, D)
        ctx.sink_dtype = sink.dtype
        return torch.addmm(out, s2, bd)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        bd, s2 = ctx.saved_tensors
        H, D = ctx.hd
        need = ctx.needs_input_grad
        g = grad.contiguous()
        ds = (g @ bd.t()).to(ctx.sink_dtype) if need[2] else None
        dt = None
        if need[1]:
            full = (s2.t() @ g).view(H, H, D)
            idx = torch.arange(H, device=g.device)
            dt = full[idx, idx]
        return (grad if need[0] else None), dt, ds
