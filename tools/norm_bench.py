"""RMS norm kernels (csrc/mmfs_norm.hip) alone: forward and backward at the LLM block's shapes."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
from mmfs_amd.functions.norm_func import RMSNormFunction

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for rows, C in ((8192, 4096), (2048, 4096), (5376, 1024), (512, 4096)):
    x = torch.randn(rows, C, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    w = torch.randn(C, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    go = torch.randn(rows, C, device="cuda", dtype=torch.bfloat16)
    y = RMSNormFunction.apply(x, w, 1e-6)
    fwd = timeit(lambda: RMSNormFunction.apply(x, w, 1e-6))
    def bwd():
        torch.autograd.grad(y, (x, w), go, retain_graph=True)
    t = timeit(bwd)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            bwd()
        torch.cuda.synchronize()
    dev = {e.key[:60]: e.device_time_total / 5 for e in prof.key_averages() if e.device_time_total > 0}
    mb = rows * C * 2 / 1e6
    print("rows %5d C %4d: forward %6.1f us (%.0f GB/s)  backward %6.1f us (%.0f GB/s of 3 x %.0f MB)" % (rows, C, fwd, 2 * mb / fwd * 1e3 / 1e3 * 1e3 / 1e3, t, 3 * mb / t * 1e3, mb), flush=True)
    print("      device time per backward: " + ", ".join("%s %.1f us" % (k.split("(")[0][-40:], v) for k, v in sorted(dev.items(), key=lambda kv: -kv[1])[:4]), flush=True)
