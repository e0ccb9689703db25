"""Debugging aid: the toy MMFSNet's training step through mmfs_amd.graphed with a trace of every capture phase.
usage: python tools/debug/graphed_capture.py [f32|bf16] [global|thread_local] [shared|private]"""
import contextlib, faulthandler, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd"), os.path.join(ROOT, "tests")]
faulthandler.enable()
import numpy as np
import torch
from helpers import load_golden
from mmfs_amd import graphed
from mmfs_amd.blocks import MMFSNet

dt = {"f32": torch.float32, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
graphed.capture_error_mode = sys.argv[2] if len(sys.argv) > 2 else "thread_local"
graphed.share_pool = (sys.argv[3] if len(sys.argv) > 3 else "shared") == "shared"
graphed.trace = lambda m: print("   [graphed]", m, file=sys.stderr, flush=True)
z = load_golden("block_sd_mmfs_net")
def T(a, dtype):
    t = torch.from_numpy(np.asarray(a))
    return (t.to(dtype) if t.is_floating_point() else t).to("cuda")
with contextlib.redirect_stdout(io.StringIO()):
    net = MMFSNet(input_channel=32, block_out_channels=[16, 24], layers_per_block=2, downsample_factor=8, n_levels=3,
                  n_points=2, gradient_checkpointing=True, spatial_shapes=[64, 32, 16])
sd = {k[len("param."):]: torch.from_numpy(np.asarray(v)) for k, v in z.items() if k.startswith("param.")}
net.load_state_dict(sd, strict=False)
net = net.to("cuda", dt).train()
if len(sys.argv) > 4 and sys.argv[4] == "conv":
    with torch.no_grad():
        for blk in net._blocks():
            blk.conv.weight.normal_(0, 0.3)
for step in range(5):
    print("step", step, file=sys.stderr, flush=True)
    net.zero_grad(set_to_none=True)
    res = [T(z[f"res.{i}"], dt).requires_grad_(True) for i in range(6)]
    feats = [T(z[f"feat.{i}"], dt).requires_grad_(True) for i in range(3)]
    mid = T(z["mid"], dt).requires_grad_(True)
    m, rr = net(mid, res, feats, T(z["ms_mask"], None))
    (m.float().sum() + sum(r.float().sum() for r in rr)).backward()
    torch.cuda.synchronize()
print("ok", graphed.stats, float(mid.grad.float().abs().sum()))
