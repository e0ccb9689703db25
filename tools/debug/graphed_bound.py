"""Debugging aid: host / device time of the warm-up run of an LLM layer's training call at BASELINE config 3's geometry."""
import contextlib, io, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
from mmfs_amd import graphed
from mmfs_amd.blocks import LlamaMMFSAttention
graphed.trace = lambda m: print("   [graphed]", m, flush=True) if m.startswith(("forward:", "not", "refused")) else None
cfg = types.SimpleNamespace(hidden_size=4096, num_attention_heads=32, rms_norm_eps=1e-6, max_position_embeddings=2048,
                            image_embed_dim=1024, spatial_shapes=[32, 16, 8])
with contextlib.redirect_stdout(io.StringIO()):
    layer = LlamaMMFSAttention(cfg, 0).to("cuda", torch.bfloat16).train()
feats = torch.randn(4, 1, 1344, 1024, device="cuda", dtype=torch.bfloat16)
import time
layers = [layer] + [type(layer)(cfg, 4 * i).to("cuda", torch.bfloat16).train() for i in range(1, 8)]
for Lq in (128, 512, 2048):
    hidden = torch.randn(4, Lq, 4096, device="cuda", dtype=torch.bfloat16)
    mask = torch.ones(4, Lq, 1, device="cuda")
    def step():
        for l in layers:
            for p in l.parameters():
                p.grad = None
        x = hidden.clone().requires_grad_(True)
        h = x
        for l in layers:
            h = h + l(h, feats, mask)
        h.backward(torch.ones_like(h))
    for on in (False, True, False, True):
        graphed.enabled = on
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        print("Lq", Lq, "graphs" if on else "plain ", "%.3f ms" % ((time.perf_counter() - t0) * 100), graphed.stats, flush=True)
