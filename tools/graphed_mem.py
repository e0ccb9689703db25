"""BASELINE config 4's training step (13 blocks at 512 px, B = 8, bf16, gradient checkpointing) through the per-block graphs:
step time and device memory with the graphs' transients in ONE pool (default) or one pool per recorded call.
usage: python tools/graphed_mem.py [shared|private|off|recompute]   (recompute: the recorded calls recompute like the checkpoints
they replace instead of keeping their activations)"""
import contextlib, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
from mmfs_amd import graphed
from mmfs_amd.blocks import MMFSNet
mode = sys.argv[1] if len(sys.argv) > 1 else "shared"
graphed.enabled = mode != "off"
graphed.share_pool = mode != "private"
from mmfs_amd.blocks.sd_mmfs import MMFSBlock
if mode in ("recompute", "private"):
    MMFSBlock.graph_keeps_activations = False
dev, dt, B, n = "cuda", torch.bfloat16, 8, 1
with contextlib.redirect_stdout(io.StringIO()):
    net = MMFSNet(input_channel=1024, block_out_channels=[320, 640, 1280, 1280], layers_per_block=2,
                  n_levels=4, n_points=8, gradient_checkpointing=True, spatial_shapes=[64, 32, 16, 8]).to(dev, dt).train()
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
def step():
    for p in net.parameters():
        p.grad = None
    r = [x.clone().requires_grad_(True) for x in res]
    m, rr = net(mid.clone().requires_grad_(True), r, feats, mask)
    (m.float().sum() + sum(x.float().sum() for x in rr)).backward()
for _ in range(4):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print(json.dumps({"mode": mode, "ms_per_step": round((time.perf_counter() - t0) * 100, 3),
                  "reserved_GB": round(torch.cuda.memory_reserved() / 2**30, 2),
                  "allocated_GB": round(torch.cuda.memory_allocated() / 2**30, 2),
                  "peak_allocated_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2), "stats": graphed.stats}))
