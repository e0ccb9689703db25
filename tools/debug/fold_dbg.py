"""Is the query projection folded in the schedule's path?  (debugging aid)"""
import contextlib, io, os, sys, types
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
from mmfs_amd.blocks import LlamaMMFSAttention, LlamaMMFSSchedule
import MultiScaleDeformableAttention as MSDA
cfg = types.SimpleNamespace(hidden_size=4096, num_attention_heads=32, rms_norm_eps=1e-6,
                            max_position_embeddings=2048, image_embed_dim=1024, spatial_shapes=[32, 16, 8])
with contextlib.redirect_stdout(io.StringIO()):
    layers = [LlamaMMFSAttention(cfg, 0).to("cuda", torch.bfloat16) for _ in range(2)]
B, n, S, Lq = 4, 1, 1344, 2048
feats = torch.randn(B, n, S, 1024, device="cuda", dtype=torch.bfloat16)
hidden = torch.randn(B, Lq, 4096, device="cuda", dtype=torch.bfloat16)
mask = torch.ones(B, Lq, n, device="cuda")
sched = LlamaMMFSSchedule(layers)
def names(fn):
    log = []
    MSDA._event_log = log
    try:
        with torch.no_grad():
            fn()
    finally:
        MSDA._event_log = None
    return [n for n, _, _ in log]
for l in layers:
    l.eval()
def plain():
    h = hidden
    for l in layers:
        h = h + l(h, feats, mask)
def sch():
    bank = sched.project(feats); ranks = sched.image_ranks(mask, Lq)
    h = hidden
    for k, l in enumerate(layers):
        h = l(h, feats, mask, value=bank.values[k], image_ranks=ranks, residual=h)
for tag, fn in (("plain", plain), ("sched", sch), ("plain", plain)):
    print(tag, names(fn), [l.attn._tables is not None and l.attn._tables[1][6] is not None for l in layers])
    for l in layers:
        l.train(False)
for l in layers:
    l.train(True)
print("train mode, no_grad:", names(sch), [l.attn._tables for l in layers][0] is None)
for l in layers:
    l.train(False)
print("back in eval:", names(sch), [l.attn._tables is not None and l.attn._tables[1][6] is not None for l in layers])

# ---- the flops / times module_bench sees
import time
from torch.utils.flop_counter import FlopCounterMode
with contextlib.redirect_stdout(io.StringIO()):
    layers = [LlamaMMFSAttention(cfg, 0).to("cuda", torch.bfloat16).eval() for _ in range(8)]
sched = LlamaMMFSSchedule(layers)
def t(fn, it=10):
    with torch.no_grad():
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(it): fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3
def fl(fn):
    with torch.no_grad():
        fn()
        with FlopCounterMode(display=False) as fc:
            fn()
    return fc.get_total_flops() / 1e9
for tag, fn in (("plain", plain), ("sched kept", sch), ("plain", plain), ("sched kept", sch)):
    print(tag, "ms %.3f" % t(fn), "GFLOP %.0f" % fl(fn), "ms again %.3f" % t(fn))

# ---- after a training step
for l in layers:
    l.train(True)
h = hidden.clone().requires_grad_(True)
bank = sched.project(feats); ranks = sched.image_ranks(mask, Lq)
x = h
for k, l in enumerate(layers):
    x = l(x, feats, mask, value=bank.values[k], image_ranks=ranks, residual=x)
x.backward(torch.ones_like(x))
for l in layers:
    l.train(False)
from mmfs_amd.levels import hook_free
print("after a training step: hook_free(dom)", hook_free(layers[0].attn.dynamic_offset_mask),
      {k: bool(getattr(layers[0].attn.dynamic_offset_mask, k, None)) for k in ("_forward_hooks", "_forward_pre_hooks", "_backward_hooks", "_backward_pre_hooks")})
import torch.nn.modules.module as _m
print({k: len(getattr(_m, k)) for k in ("_global_forward_hooks", "_global_forward_pre_hooks", "_global_backward_hooks", "_global_backward_pre_hooks")})
for tag, fn in (("plain", plain), ("sched kept", sch)):
    print(tag, "ms %.3f" % t(fn), "GFLOP %.0f" % fl(fn), [l.attn._tables[1][6] is not None for l in layers][:2])
