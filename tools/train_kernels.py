"""Which kernels a TRAINING step of the MMFS stacks launches, by device time (tools/module_bench.py's cfg3 at 2048
tokens with the schedule, cfg4's MMFSNet step): python tools/train_kernels.py cfg3|cfg4"""
import contextlib, io, os, sys, types
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd"), os.path.join(ROOT, "tools")]
import torch
from torch.profiler import ProfilerActivity, profile

which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
seq = len(sys.argv) > 2 and sys.argv[2] == "seq"        # cfg3 seq: ONE layer's forward + backward, kernel by kernel in time order
dev, dt = "cuda", torch.bfloat16
if which == "cfg3":
    from mmfs_amd.blocks import LlamaMMFSAttention, LlamaMMFSSchedule
    cfg = types.SimpleNamespace(hidden_size=4096, num_attention_heads=32, rms_norm_eps=1e-6,
                                max_position_embeddings=2048, image_embed_dim=1024, spatial_shapes=[32, 16, 8])
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        layers = [LlamaMMFSAttention(cfg, 4 * i).to(dev, dt) for i in range(1 if seq else 8)]
    with torch.no_grad():
        for l in layers:
            l.gate.fill_(0.5)
            l.attn.sampling_offsets.weight.normal_(0, 0.01)
    B, n, S, Lq = 4, 1, 1344, 2048
    feats = torch.randn(B, n, S, 1024, device=dev, dtype=dt)
    hidden = torch.randn(B, Lq, 4096, device=dev, dtype=dt)
    mask = torch.ones(B, Lq, n, device=dev)
    sched = LlamaMMFSSchedule(layers)

    def step():
        for l in layers:
            for p in l.parameters():
                p.grad = None
        h = hidden.clone().requires_grad_(True)
        bank = sched.project(feats)
        ranks = sched.image_ranks(mask, Lq)
        x = h
        for k, l in enumerate(layers):
            x = l(x, feats, mask, value=bank.values[k], image_ranks=ranks, residual=x)
        x.backward(torch.ones_like(x))
else:
    from mmfs_amd.blocks import MMFSNet
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
    net.train()
    if "nockpt" in sys.argv:            # (the replayed step of tools/module_bench.py runs without checkpointing)
        for blk in net._blocks():
            blk.gradient_checkpointing = False

    def step():
        for p in net.parameters():
            p.grad = None
        r = [x.clone().requires_grad_(True) for x in res]
        m, rr = net(mid.clone().requires_grad_(True), r, feats, mask)
        (m.float().sum() + sum(x.float().sum() for x in rr)).backward()

for _ in range(3):
    step()
torch.cuda.synchronize()
iters = 3
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows) / iters
print("%s: %.0f us of kernels per step, %d launches" % (which, tot, sum(e.count for e in rows) // iters))
for e in rows[:45]:
    print("  %8.1f us  x%-4d %s" % (e.device_time_total / iters, e.count // iters, e.key[:140]))

if seq:
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        step()
        torch.cuda.synchronize()
    evs = sorted((e for e in prof.events() if e.device_type.name == "CUDA"), key=lambda e: e.time_range.start)
    print("one layer, forward + backward, in time order: %d kernels" % len(evs))
    for e in evs:
        print("  %7.1f us  %s" % (e.device_time, e.name[:150]))
