"""Which statement of a Linear layer's weight gradient the BLAS library runs faster at the LLM layers' shapes:
dW = g^T x (what autograd's Linear backward issues) or dW^T = x^T g -- python tools/gemm_forms.py"""
import time, torch
dev, dt = "cuda", torch.bfloat16
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6
for T, N, K in ((8192, 4096, 4096), (8192, 4096, 1024), (8192, 640, 4096), (2048, 4096, 4096), (512, 4096, 4096),
                (32768, 1280, 1280), (32768, 320, 320)):
    g = torch.randn(T, N, device=dev, dtype=dt); x = torch.randn(T, K, device=dev, dtype=dt); w = torch.randn(N, K, device=dev, dtype=dt)
    fl = 2 * T * N * K
    a = t(lambda: g.t() @ x)
    b = t(lambda: x.t() @ g)
    c = t(lambda: torch.mm(g.t().contiguous(), x))
    f = t(lambda: x @ w.t())
    dx = t(lambda: g @ w)
    print("T=%-6d N=%-5d K=%-5d  fwd x W^T %7.1f us (%.2f PF/s)  dx g W %7.1f (%.2f)  dW g^T x %7.1f (%.2f)  x^T g %7.1f (%.2f)  g^T.contiguous() x %7.1f"
          % (T, N, K, f, fl / f / 1e9, dx, fl / dx / 1e9, a, fl / a / 1e9, b, fl / b / 1e9, c))
