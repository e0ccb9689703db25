"""Debugging aid: the toy MMFSNet's checkpointed training step, per-block graphs against the plain eager path, bit for bit (bf16)."""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from helpers import load_golden
from mmfs_amd import graphed
from mmfs_amd.blocks import MMFSNet
dt = torch.bfloat16
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
torch.manual_seed(1)
with torch.no_grad():
    for blk in net._blocks():
        blk.conv.weight.normal_(0, 0.3)
feats_req = len(sys.argv) > 1 and sys.argv[1] == "featgrad"
def step(scale):
    net.zero_grad(set_to_none=True)
    res = [(T(z[f"res.{i}"], dt) * scale).requires_grad_(True) for i in range(6)]
    feats = [T(z[f"feat.{i}"], dt).requires_grad_(feats_req) for i in range(3)]
    mid = (T(z["mid"], dt) * scale).requires_grad_(True)
    m, rr = net(mid, res, feats, T(z["ms_mask"], None))
    g = torch.Generator().manual_seed(5)
    loss = (m.float() * torch.randn(m.shape, generator=g).to("cuda")).sum()
    for r in rr:
        loss = loss + (r.float() * torch.randn(r.shape, generator=g).to("cuda")).sum()
    loss.backward()
    out = {"out.mid": m.detach(), "g.mid": mid.grad}
    out.update({f"out.res{i}": r.detach() for i, r in enumerate(rr)})
    out.update({f"g.res{i}": r.grad for i, r in enumerate(res)})
    if feats_req:
        out.update({f"g.feat{i}": f.grad for i, f in enumerate(feats)})
    out.update({"p." + k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    return out
for _ in range(3):
    step(1.0)
a = step(0.7)
a2 = step(0.7)
graphed.enabled = False
b = step(0.7)
b2 = step(0.7)
print("stats", graphed.stats)
for name, (x, y) in (("graphs vs eager", (a, b)), ("graphs vs graphs", (a, a2)), ("eager vs eager", (b, b2))):
    bad = [(k, float((x[k].float() - y[k].float()).abs().max()), float(y[k].float().abs().max())) for k in y if not torch.equal(x[k], y[k])]
    print(name, ":", len(bad), "of", len(y), "tensors differ", bad[:12])
