"""Debugging aid: which recorded call leaves the default CUDA generator in capture mode?"""
import contextlib, faulthandler, gc, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd"), os.path.join(ROOT, "tests")]
faulthandler.enable()
import numpy as np
import torch
from helpers import load_golden
from mmfs_amd import graphed
from mmfs_amd.blocks import MMFSNet

def probe(tag):
    try:
        torch.empty(4, device="cuda").normal_()
        ok = "rng ok"
    except Exception as e:
        ok = "RNG STUCK: " + str(e)[:60]
    print("   [probe]", tag, ok, "gc", gc.get_count(), file=sys.stderr, flush=True)

def trace(m):
    if m in ("forward recorded", "backward recorded", "warm-up done"):
        probe(m)
graphed.trace = trace
z = load_golden("block_sd_mmfs_net")
def T(a, dtype):
    t = torch.from_numpy(np.asarray(a))
    return (t.to(dtype) if t.is_floating_point() else t).to("cuda")
def run(once):
    with contextlib.redirect_stdout(io.StringIO()):
        net = MMFSNet(input_channel=32, block_out_channels=[16, 24], layers_per_block=2, downsample_factor=8, n_levels=3,
                      n_points=2, gradient_checkpointing=True, spatial_shapes=[64, 32, 16])
    sd = {k[len("param."):]: torch.from_numpy(np.asarray(v)) for k, v in z.items() if k.startswith("param.")}
    net.load_state_dict(sd, strict=False)
    net = net.to("cuda").train()
    net.project_once_in_training = once
    for step in range(4):
        print("once", once, "step", step, file=sys.stderr, flush=True)
        net.zero_grad(set_to_none=True)
        res = [T(z[f"res.{i}"], torch.float32).requires_grad_(True) for i in range(6)]
        feats = [T(z[f"feat.{i}"], torch.float32).requires_grad_(True) for i in range(3)]
        mid = T(z["mid"], torch.float32).requires_grad_(True)
        m, rr = net(mid, res, feats, T(z["ms_mask"], None))
        (m.float().sum() + sum(r.float().sum() for r in rr)).backward()
        torch.cuda.synchronize()
        probe("after step")
for once in [bool(int(a)) for a in (sys.argv[1:] or ["1", "0"])]:
    run(once)
    probe("after run")
print("done", graphed.stats)
