#!/usr/bin/env python3
"""the 13-block MMFSNet schedule at 512 px (UNet latent 64x64), B=8, bf16,
4 levels of one image, random residuals and features.  Times
  sampling  forward under no_grad -- the reference's schedule (13 LayerNorms + 13 value GEMMs per
            call), the fused schedule (one normalisation), and the fused schedule with the projected
            bank kept across denoising steps;
  training  forward + backward with gradient checkpointing as the reference builds it.
    python tools/net_bench.py [B] [n_images]
Not the contract benchmark (bench.py)."""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch  # noqa: E402
from mmfs_amd.blocks import MMFSNet  # noqa: E402

dev, dt = "cuda", torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
with contextlib.redirect_stdout(io.StringIO()):
    net = MMFSNet(input_channel=1024, block_out_channels=[320, 640, 1280, 1280], layers_per_block=2,
                  n_levels=4, n_points=8, gradient_checkpointing=True, spatial_shapes=[64, 32, 16, 8]).to(dev, dt)
torch.manual_seed(0)
with torch.no_grad():
    for blk in net._blocks():
        blk.conv.weight.normal_(0, 0.02)
        blk.mmfs.sampling_offsets.weight.normal_(0, 0.01)
        blk.feat_norm.weight.uniform_(0.5, 1.5)
        blk.feat_norm.bias.normal_(0, 0.1)
# the UNet's residuals at 512 px: (channels, side) per down residual, then the mid sample
geom = list(zip([320] * 4 + [640] * 3 + [1280] * 5, [64] * 3 + [32] * 3 + [16] * 3 + [8] * 3))
res = [torch.randn(B, c, s, s, device=dev, dtype=dt) for c, s in geom]
mid = torch.randn(B, 1280, 8, 8, device=dev, dtype=dt)
feats = [torch.randn(B, n, 1024, s, s, device=dev, dtype=dt) for s in (64, 32, 16, 8)]
mask = torch.ones(B, n, device=dev, dtype=torch.long)


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


def sample_step():
    with torch.no_grad():
        return net(mid, res, feats, mask)


net.eval()
rows = []
net.fused_schedule, net.cache_projected_features = False, False
want = sample_step()
rows.append(("sampling step, reference schedule", timed(sample_step)))
net.fused_schedule = True
got = sample_step()
err = max(float((a.float() - b.float()).abs().max()) for a, b in zip((got[0],) + got[1], (want[0],) + want[1]))
ref = max(float(a.float().abs().max()) for a in (want[0],) + want[1])
rows.append(("sampling step, one normalisation", timed(sample_step)))
net.cache_projected_features = True
rows.append(("sampling step, projected bank kept", timed(sample_step)))
from mmfs_amd.graphs import GraphedMMFSNet  # noqa: E402
graphed = GraphedMMFSNet(net, mid, res, feats, mask)
eager, replay = sample_step(), graphed(mid, res)
same = all(torch.equal(a, b) for a, b in zip((eager[0],) + tuple(eager[1]), (replay[0],) + tuple(replay[1])))
rows.append((f"sampling step, HIP graph replay (bit-equal: {same})", timed(lambda: graphed(mid, res))))
del graphed, replay
net.clear_feature_cache()

net.train()
for r in res + [mid] + feats:
    r.requires_grad_(True)


def train_step():
    m, rs = net(mid, res, feats, mask)
    loss = m.float().sum() + sum(r.float().sum() for r in rs)
    loss.backward()


net.fused_schedule = False
rows.append(("training fwd+bwd, reference schedule", timed(train_step, 5, 2)))
net.fused_schedule = True
rows.append(("training fwd+bwd, one normalisation", timed(train_step, 5, 2)))
print(f"MMFSNet config 4: B={B} n={n} bf16, 13 blocks; fused vs reference schedule max abs diff {err:.3e} (max |out| {ref:.2f})")
for k, v in rows:
    print(f"  {k:52s} {v:8.2f} ms")
print(f"  peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")

if os.environ.get("NET_PROFILE"):
    from torch.profiler import profile, ProfilerActivity
    mode = os.environ["NET_PROFILE"]                # ref | fused | cache | train
    net.train(mode == "train")
    net.fused_schedule = mode != "ref"
    net.cache_projected_features = mode == "cache"
    step = train_step if mode == "train" else sample_step
    step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=70))
