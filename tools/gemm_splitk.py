"""A Linear layer's weight gradient dW = g^T x when tokens >> features (the image decoder's query-side layers: 32768 tokens,
320 x 320 weights): the BLAS call reduces over all tokens inside one workgroup per output tile; the same product as a batch of
S shorter ones + a sum -- python tools/gemm_splitk.py"""
import time, torch
dev, dt = "cuda", torch.bfloat16
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6
for T, N, K in ((32768, 320, 320), (32768, 640, 320), (32768, 320, 1024), (32768, 1280, 1280), (8192, 640, 640), (8192, 1280, 640),
                (8192, 640, 1024), (2048, 1280, 1280), (8192, 640, 4096), (8192, 4096, 1024)):
    g = torch.randn(T, N, device=dev, dtype=dt); x = torch.randn(T, K, device=dev, dtype=dt)
    fl = 2 * T * N * K
    base = t(lambda: g.t() @ x)
    out = ["T=%-6d N=%-5d K=%-5d  g^T x %7.1f us (%.2f PF/s)" % (T, N, K, base, fl / base / 1e9)]
    for S in (4, 8, 16, 32, 64):
        if T % S: continue
        f = lambda: torch.bmm(g.view(S, T // S, N).transpose(1, 2), x.view(S, T // S, K)).sum(0)
        f32 = lambda: torch.bmm(g.view(S, T // S, N).transpose(1, 2), x.view(S, T // S, K)).sum(0, dtype=torch.float32).to(dt)
        out.append("S=%d: %.1f / %.1f" % (S, t(f), t(f32)))
    print("  ".join(out))
