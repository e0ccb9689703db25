"""Which kernels one denoising step of MMFSNet (13 blocks at 512 px, B = 8; tools/module_bench.py's cfg4) launches."""
import contextlib, io, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
from torch.profiler import ProfilerActivity, profile
from mmfs_amd.blocks import MMFSNet

dev, dt = "cuda", torch.bfloat16
B, n = 8, 1
with contextlib.redirect_stdout(io.StringIO()):
    net = MMFSNet(input_channel=1024, block_out_channels=[320, 640, 1280, 1280], layers_per_block=2,
                  n_levels=4, n_points=8, gradient_checkpointing=True, spatial_shapes=[64, 32, 16, 8]).to(dev, dt)
torch.manual_seed(0)
with torch.no_grad():
    for blk in net._blocks():
        blk.conv.weight.normal_(0, 0.02)
        blk.mmfs.sampling_offsets.weight.normal_(0, 0.01)
geom = list(zip([320] * 4 + [640] * 3 + [1280] * 5, [64] * 3 + [32] * 3 + [16] * 3 + [8] * 3))
res = [torch.randn(B, c, s, s, device=dev, dtype=dt) for c, s in geom]
mid = torch.randn(B, 1280, 8, 8, device=dev, dtype=dt)
feats = [torch.randn(B, n, 1024, s, s, device=dev, dtype=dt) for s in (64, 32, 16, 8)]
mask = torch.ones(B, n, device=dev, dtype=torch.long)
net.eval()


def step():
    with torch.no_grad():
        return net(mid, res, feats, mask)


for _ in range(5):
    step()
torch.cuda.synchronize()
iters = 5
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows) / iters
print("sampling step: %.0f us of kernels, %d launches" % (tot, sum(e.count for e in rows) // iters))
for e in rows[:32]:
    print("  %8.1f us  x%-4d %s" % (e.device_time_total / iters, e.count // iters, e.key[:150]))
