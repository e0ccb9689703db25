#!/usr/bin/env python3
"""What does it cost a TRAINING step that the sampling plan hands `loc` / `attn` to the op as tensors (written once,
read by the forward, by the taps kernel and by the sort), instead of being recomputed inside the sampler as the
inference path does (csrc/mmfs_plan.hip, mmfs_sample_fwd)?  VERDICT r2, next-round item 6: "prove with a measurement
that the 3 * pts round trip is < 3 % of an SD block step".

Measures, per geometry: the MMFS module's forward + backward step, and the time of moving 3 * pts elements of the
storage type once out (a write) and three times in (reads) at the rates this GPU gives plain copies of that size."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
from mmfs_amd.modules import MMFS
from mmfs_amd.levels import make_level_tables

dev, dt = "cuda", torch.bfloat16
CASES = {
    "sd_block_320 (B=8, 64x64 queries)": dict(mod=dict(d_model=1024, d_query=320, d_value=1024, d_out=320, n_levels=4, n_heads=16, n_points=8,
                                              ratio=1.0, offset_init_magnitude=1, spatial_shapes=[64, 32, 16, 8], base_spatial_shape=64,
                                              max_num_image_per_seq=10), B=8, Lq=4096, n=1, shapes=[(64, 64), (32, 32), (16, 16), (8, 8)]),
    "llm_layer_7b (B=4, 2048 tokens, 4 images)": dict(mod=dict(d_model=4096, d_query=4096, d_value=1024, d_out=4096, n_levels=3, n_heads=16, n_points=8,
                                                      ratio=0.25, offset_init_magnitude=3.0, spatial_shapes=[32, 16, 8], base_spatial_shape=16,
                                                      max_num_image_per_seq=50), B=4, Lq=2048, n=4, shapes=[(32, 32), (16, 16), (8, 8)]),
}


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, c in CASES.items():
    with contextlib.redirect_stdout(io.StringIO()):
        m = MMFS(**c["mod"]).to(dev, dt)
    with torch.no_grad():
        m.sampling_offsets.weight.normal_(0, 0.01)
    B, Lq, n = c["B"], c["Lq"], c["n"]
    sh, st, S = make_level_tables(c["shapes"], n, dev)
    q = torch.randn(B, Lq, c["mod"]["d_query"], device=dev, dtype=dt, requires_grad=True)
    f = torch.randn(B, n, S // n, c["mod"]["d_value"], device=dev, dtype=dt, requires_grad=True)
    ref = torch.full((1, Lq, 1, 2), 0.5, device=dev, dtype=dt)
    mask = torch.ones(B, n, device=dev, dtype=torch.long)

    def step():
        out = m(q, ref, f, sh, st, None, mask)
        out.backward(torch.ones_like(out))

    ms_step = timed(step)
    pts = B * Lq * c["mod"]["n_heads"] * c["mod"]["n_levels"] * n * c["mod"]["n_points"]
    a = torch.empty(3 * pts, device=dev, dtype=dt).normal_()
    b = torch.empty_like(a)
    ms_copy = timed(lambda: b.copy_(a), n=50)                # one read + one write of 3 * pts elements
    ms_read = timed(lambda: a.sum(), n=50)                   # one read
    trip = ms_copy + 2 * ms_read                             # written once, read three times
    print(f"{name}: module fwd+bwd {ms_step:.3f} ms; loc+attn = {3 * pts * 2 / 1e6:.0f} MB; copy {ms_copy * 1e3:.1f} us, "
          f"read {ms_read * 1e3:.1f} us -> round trip (1 write + 3 reads) {trip * 1e3:.1f} us = {100 * trip / ms_step:.2f} % of the step")
