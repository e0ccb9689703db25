#!/usr/bin/env python3
"""Times the MMFS module (projections + sampling plan + op) forward and forward+backward, with a
torch-profiler breakdown by kernel, at the reference-real geometries.
Not the contract benchmark (that is bench.py); this shows what surrounds the op."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
from mmfs_amd.modules import MMFS
from mmfs_amd.levels import make_level_tables

dev = "cuda"
CASES = {
    # SD block at the 64x64 UNet resolution: d_query=320, 4 levels of one image
    "sd_320": dict(mod=dict(d_model=1024, d_query=320, d_value=1024, d_out=320, n_levels=4, n_heads=16, n_points=8,
                            ratio=1.0, offset_init_magnitude=1, spatial_shapes=[64, 32, 16, 8], base_spatial_shape=64,
                            max_num_image_per_seq=10), B=8, Lq=4096, n=1, shapes=[(64, 64), (32, 32), (16, 16), (8, 8)]),
    # LLM layer, Vicuna-7B geometry, 1 image, 2048 tokens
    "llm_7b": dict(mod=dict(d_model=4096, d_query=4096, d_value=1024, d_out=4096, n_levels=3, n_heads=16, n_points=8,
                            ratio=0.25, offset_init_magnitude=3.0, spatial_shapes=[32, 16, 8], base_spatial_shape=16,
                            max_num_image_per_seq=50), B=4, Lq=2048, n=1, shapes=[(32, 32), (16, 16), (8, 8)]),
    "llm_7b_n4": dict(mod=dict(d_model=4096, d_query=4096, d_value=1024, d_out=4096, n_levels=3, n_heads=16, n_points=8,
                               ratio=0.25, offset_init_magnitude=3.0, spatial_shapes=[32, 16, 8], base_spatial_shape=16,
                               max_num_image_per_seq=50), B=4, Lq=2048, n=4, shapes=[(32, 32), (16, 16), (8, 8)]),
}
dt = torch.bfloat16
for name, c in CASES.items():
    with contextlib.redirect_stdout(io.StringIO()):
        m = MMFS(**c["mod"]).to(dev, dt)
    with torch.no_grad():
        m.sampling_offsets.weight.normal_(0, 0.01)
    B, Lq, n = c["B"], c["Lq"], c["n"]
    sh, st, S = make_level_tables(c["shapes"], n, dev)
    hw = S // n
    q = torch.randn(B, Lq, c["mod"]["d_query"], device=dev, dtype=dt, requires_grad=True)
    f = torch.randn(B, n, hw, c["mod"]["d_value"], device=dev, dtype=dt, requires_grad=True)
    ref = torch.full((1, Lq, 1, 2), 0.5, device=dev)
    mask = torch.ones(B, n, device=dev, dtype=torch.long)
    def fwd():
        return m(q, ref.to(dt), f, sh, st, None, mask)
    def fwdbwd():
        out = fwd()
        out.backward(torch.ones_like(out))
    for fn, label in ((fwd, "fwd"), (fwdbwd, "fwd+bwd")):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize()
        print(f"{name:10s} {label:8s} {(time.perf_counter() - t0) / 20 * 1e3:8.3f} ms")
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5): fwdbwd()
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:12]
    tot = sum(e.device_time_total for e in prof.key_averages())
    for e in rows:
        print(f"      {e.device_time_total / 5:9.1f} us  {100 * e.device_time_total / tot:5.1f}%  x{e.count // 5:<3d} {e.key[:90]}")
