"""The op under HIP-graph capture (runs last: a broken capture must not hide other failures).

Small shapes are bound by the host's launch work (tools/host_floor.py); replayed from a graph a forward +
backward costs less than half (tools/graph_step.py, profiles/r02ak_graph_step.jsonl).  That only works while
the op stays capture-safe: no device->host copy per call, every launch on torch's current stream,
workspaces from the caching allocator.  This test captures a step and holds the replayed outputs to the
eager ones."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    "cfg1_small": dict(B=2, Nq=256, H=8, D=32, P=4, dt=torch.float32, shapes=[(64, 64), (32, 32), (16, 16), (8, 8)]),
    "decode": dict(B=4, Nq=1, H=16, D=64, P=8, dt=torch.bfloat16, shapes=[(32, 32), (16, 16), (8, 8)]),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_forward_backward_replays_from_a_graph(name):
    from mmfs_amd.functions import MSDeformAttnFunction
    from mmfs_amd.levels import make_level_tables
    w, dev = CASES[name], "cuda"
    sh, st, _ = make_level_tables(w["shapes"], 1, dev)
    S, L = sum(h * ww for h, ww in w["shapes"]), len(w["shapes"])
    g = torch.Generator(device=dev).manual_seed(3)
    B, Nq, H, D, P, dt = w["B"], w["Nq"], w["H"], w["D"], w["P"], w["dt"]
    value = torch.rand(B, S, H, D, device=dev, generator=g).to(dt).requires_grad_(True)
    loc = torch.rand(B, Nq, H, L, P, 2, device=dev, generator=g).to(dt).requires_grad_(True)
    attn = torch.rand(B, Nq, H, L, P, device=dev, generator=g)
    attn = (attn / attn.sum((-1, -2), keepdim=True)).to(dt).requires_grad_(True)
    grad = torch.randn(B, Nq, H * D, device=dev, generator=g).to(dt)

    def step():
        out = MSDeformAttnFunction.apply(value, sh, st, loc, attn, 64)
        return (out,) + torch.autograd.grad(out, (value, loc, attn), grad)

    want = [t.detach().clone() for t in step()]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up on a side stream, as capture asks
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    # (the order of a cell's records, hence of the fp32 sums of grad_value, is not fixed from run to run)
    tol = dict(rtol=1e-4, atol=1e-5) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    for n, a, b in zip(("out", "grad_value", "grad_loc", "grad_attn"), captured, want):
        assert torch.isfinite(a).all(), n
        assert torch.allclose(a.float(), b.float(), **tol), (n, float((a.float() - b.float()).abs().max()))
