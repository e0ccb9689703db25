#!/usr/bin/env python3
"""Module-level measurement lines for BASELINE configs 3 and 4 -- not the contract
benchmark (that is bench.py, the op at config 2), but what surrounds the op on the two decoder paths:

  cfg3  text forward, Vicuna-7B geometry: the 8 MMFS layers of the LLM (LlamaMMFSAttention: RMSNorm x2,
        MMFS with d_query = 4096, H = 16, P = 8, L = 3 (32^2, 16^2, 8^2; S = 1344), tanh gate), B = 4, one
        image, Lq in {1 (decode), 128, 512, 2048}; the dense LLM between them is out of scope
  cfg4  image forward at 512 px: the 13-block MMFSNet schedule (sd_mmfs.py:230-272), B = 8, 4 levels of one
        image, random residuals and features; one denoising step (no_grad, eager and HIP-graph replay) and
        one training step

One JSON line per case: ms per call, the time split by kernel family from the torch profiler (GEMM =
hipBLASLt / rocBLAS kernels of the F.linear layers; op = this library's HIP kernels; other = framework
elementwise / norm / copy kernels), and the two rooflines the north star asks for:
  gemm_mfma_util   = FLOPs of the matrix products the call actually EXECUTED (counted with torch's FLOP formulas under a bare dispatch mode on
                     one call: a projection that is kept or folded away is not counted -- round 3 divided a fixed analytic
                     count by the measured time and reported utilisations above 1) / GEMM kernel time /
                     2.5 PFLOP/s (dense bf16 MFMA peak).  Products that run in this library's own kernels (the
                     small-token Linear kernel) are in the "op" family on both sides: neither FLOPs nor time here
  op_hbm_frac      = algorithmic bytes of the sampling op (SURVEY 8d formula) / op kernel time / 8 TB/s
Random weights (no checkpoints offline), bf16.   usage: python tools/module_bench.py [cfg3] [cfg4]
"""
import contextlib
import io
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

dev, dt = "cuda", torch.bfloat16
MFMA_PEAK, HBM_PEAK = 2.5e15, 8e12


def family(name):
    n = name.lower()
    if "mmfs" in n or "msda" in n:
        return "op"
    if "cijk" in n or "gemm" in n or "hipblaslt" in n or "tensile" in n or "xdl" in n:
        return "gemm"
    return "other"


def timed(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def split(fn, iters=5):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
    fam = {"op": 0.0, "gemm": 0.0, "other": 0.0}
    launches = 0
    for e in prof.key_averages():
        fam[family(e.key)] += e.device_time_total / iters
        launches += e.count / iters
    return {k: round(v, 1) for k, v in fam.items()}, int(launches)


def executed_gemm_flops(fn):
    """FLOPs of the aten matrix products (mm / addmm / bmm / convolution, forward and backward) one call of fn dispatches.
    torch's own formulas (``flop_registry``) under a bare dispatch mode: ``FlopCounterMode`` also registers process-wide
    module hooks, and a module that sees hooks does not take its folds -- the counted call would not be the timed one."""
    from torch.utils._python_dispatch import TorchDispatchMode
    from torch.utils.flop_counter import flop_registry

    class Count(TorchDispatchMode):
        total = 0

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            f = flop_registry.get(getattr(func, "_overloadpacket", None))
            if f is not None:
                self.total += int(f(*args, **(kwargs or {}), out_val=out))
            return out

    with Count() as c:
        fn()
    torch.cuda.synchronize()
    return int(c.total)


def mmfs_linear_flops(tokens, bank_tokens, d_query, d_value, d_inner, d_out, H, L, P, max_img):
    """Forward FLOPs of the five Linear layers of MMFS.forward as this build evaluates them (one
    dynamic_offset_mask GEMM on the un-repeated query; heads on the query + small table GEMMs)."""
    f = 2 * bank_tokens * d_value * d_inner                      # value_proj
    f += 2 * tokens * d_query * d_query                          # dynamic_offset_mask
    f += 2 * tokens * d_query * (H * P * 2)                      # sampling_offsets
    f += 2 * tokens * d_query * (H * L * P)                      # attention_weights (point columns)
    f += 2 * tokens * d_inner * d_out                            # output_proj
    f += 2 * max_img * d_query * (H * P * 2 + 2 * H * L * (P + 1))   # relative-position tables
    return f


def op_bytes(B, Nq, H, D, Leff, P, S, e=2, backward=False):
    pts, C = B * Nq * H * Leff * P, H * D
    fwd = e * (B * S * C + 3 * pts + B * Nq * C)
    bwd = e * (2 * B * S * C + 6 * pts + B * Nq * C)
    return fwd + (bwd if backward else 0)


def cfg3():
    from mmfs_amd.blocks import LlamaMMFSAttention, LlamaMMFSSchedule
    from mmfs_amd.graphs import GraphedLlamaMMFSStack
    cfg = types.SimpleNamespace(hidden_size=4096, num_attention_heads=32, rms_norm_eps=1e-6,
                                max_position_embeddings=2048, image_embed_dim=1024, spatial_shapes=[32, 16, 8])
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        layers = [LlamaMMFSAttention(cfg, 4 * i).to(dev, dt) for i in range(8)]
    with torch.no_grad():
        for l in layers:
            l.gate.fill_(0.5)
            l.attn.sampling_offsets.weight.normal_(0, 0.01)
    B, n, S = 4, 1, 1344
    feats = torch.randn(B, n, S, 1024, device=dev, dtype=dt)
    for Lq in (1, 128, 512, 2048):
        hidden = torch.randn(B, Lq, 4096, device=dev, dtype=dt)
        mask = torch.ones(B, Lq, n, device=dev)

        def fwd():
            with torch.no_grad():
                h = hidden
                for l in layers:
                    h = h + l(h, feats, mask)
                return h

        params = [p for l in layers for p in l.parameters()]

        def zero_grad():                # (what optimizer.zero_grad(set_to_none=True) does: no accumulation kernels in the step)
            for p in params:
                p.grad = None

        def train():
            zero_grad()
            h = hidden.clone().requires_grad_(True)
            x = h
            for l in layers:
                x = x + l(x, feats, mask)
            x.backward(torch.ones_like(x))

        sched = LlamaMMFSSchedule(layers)

        def fwd_sched(keep):
            # one RMS pass + one batched projection for all 8 layers; keep: the bank does not change between calls
            # (decode steps of a generation loop) and the projections are reused
            with torch.no_grad():
                if not keep:
                    sched.clear_cache()
                bank = sched.project(feats)
                ranks = sched.image_ranks(mask, Lq)
                h = hidden
                for k, l in enumerate(layers):
                    h = l(h, feats, mask, value=bank.values[k], image_ranks=ranks, residual=h)
                return h

        def train_sched():
            zero_grad()
            h = hidden.clone().requires_grad_(True)
            bank = sched.project(feats)
            ranks = sched.image_ranks(mask, Lq)
            x = h
            for k, l in enumerate(layers):
                x = l(x, feats, mask, value=bank.values[k], image_ranks=ranks, residual=x)
            x.backward(torch.ones_like(x))

        for l in layers:
            l.eval()
        graphed = GraphedLlamaMMFSStack(layers, hidden, feats, mask)

        def stack_fn(h):
            bank = sched.project(feats)
            ranks = sched.image_ranks(mask, Lq)
            x = h
            for k, l in enumerate(layers):
                x = l(x, feats, mask, value=bank.values[k], image_ranks=ranks, residual=x)
            return x
        gstep = [None]

        def train_graphed():
            if gstep[0] is None:                   # (captured in training mode, once per shape)
                from mmfs_amd.graphs import GraphedTrainingStep
                gstep[0] = GraphedTrainingStep(stack_fn, [hidden.clone().requires_grad_(True)], [torch.ones_like(hidden)],
                                               [p for l in layers for p in l.parameters()])
            gstep[0]([hidden], [torch.ones_like(hidden)])
        def train_plain():
            from mmfs_amd import graphed
            graphed.enabled = False
            try:
                train()
            finally:
                graphed.enabled = True
        cases = [("forward", fwd, False), ("forward, shared normalisation + batched value projection", lambda: fwd_sched(False), False),
                 ("forward, projected bank kept across calls (decode / generation)", lambda: fwd_sched(True), False),
                 ("forward, projected bank kept, HIP-graph replay", lambda: graphed(hidden), False),      # (replays what the line above runs)
                 ("forward+backward, every launch issued by the host (mmfs_amd.graphed.enabled = False: rounds 1-4)", train_plain, True),
                 ("forward+backward (round 5 default: a layer's call replays two HIP graphs from its third identical call on)", train, True),
                 ("forward+backward, shared normalisation + batched value projection, residual handed to the layer", train_sched, True),
                 ("forward+backward, whole step replayed as ONE HIP graph", train_graphed, True)]
        for label, fn, bwd in cases:
            if bwd and Lq == 1:
                continue
            for l in layers:            # (no-grad lines in eval mode -- folds and kept tables --, training lines in training mode)
                l.train(bwd)
            ms = timed(fn)
            fam, launches = split(fn)
            analytic = 8 * mmfs_linear_flops(B * Lq, B * n * S, 4096, 1024, 1024, 4096, 16, 3, 8, 50) * (3 if bwd else 1)
            counted = (lambda: fwd_sched(True)) if "graph replay" in label else train_sched if "ONE HIP graph" in label else fn
            counted()                   # (the mode change above dropped what the modules keep: count a call that has it again)
            flops = executed_gemm_flops(counted)
            ob = 8 * op_bytes(B, Lq, 16, 64, 3 * n, 8, S * n, backward=bwd)
            print(json.dumps({
                "config": "cfg3", "what": f"8 MMFS layers (Vicuna-7B geometry), B={B}, Lq={Lq}, n_images={n}, bf16, {label}",
                "ms": round(ms, 3), "kernel_us": fam, "launches": launches,
                "gemm_flops": flops, "gemm_flops_reference_schedule": analytic,
                "gemm_mfma_util": round(flops / (fam["gemm"] * 1e-6) / MFMA_PEAK, 4) if fam["gemm"] else None,
                "op_algorithmic_bytes": ob, "op_hbm_frac": round(ob / (fam["op"] * 1e-6) / HBM_PEAK, 4) if fam["op"] else None,
            }), flush=True)


def cfg4():
    from mmfs_amd.blocks import MMFSNet
    from mmfs_amd.graphs import GraphedMMFSNet
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
    S = 5440
    tokens = [(c, s * s) for c, s in geom] + [(1280, 64)]
    flops = sum(mmfs_linear_flops(B * t, 0, c, 1024, 1024, c, 16, 4, 8, 10) + 2 * B * t * c * c for c, t in tokens)
    flops_proj = 13 * 2 * B * n * S * 1024 * 1024
    ob = sum(op_bytes(B, t, 16, 64, 4 * n, 8, S * n) for _, t in tokens)

    net.eval()

    def sample_step():
        with torch.no_grad():
            return net(mid, res, feats, mask)

    sample_step()
    graphed = GraphedMMFSNet(net, mid, res, feats, mask)
    cases = [("one denoising step, eager (projected bank kept across steps)", sample_step, flops, False),
             ("one denoising step, HIP-graph replay", lambda: graphed(mid, res), flops, False)]
    net_t = net

    def train_step():
        # (the mode is set once per line, below: every train() / eval() drops what the modules keep -- the blocks' position
        # tables among it, 2.4 ms of bicubic resize each -- and a training loop does not change mode per step)
        r = [x.clone().requires_grad_(True) for x in res]
        m, rr = net_t(mid.clone().requires_grad_(True), r, feats, mask)
        (m.float().sum() + sum(x.float().sum() for x in rr)).backward()

    def train_step_plain():
        from mmfs_amd import graphed
        graphed.enabled = False
        try:
            train_step()
        finally:
            graphed.enabled = True
    cases.append(("training step, every launch issued by the host (gradient checkpointing; bank projected once; mmfs_amd.graphed.enabled = False: round 4's default)",
                  train_step_plain, 3 * (flops + flops_proj) + flops + flops_proj, True))
    cases.append(("training step (forward + backward, gradient checkpointing; bank projected once for all blocks; round 5 default: every block's "
                  "checkpointed call replays two HIP graphs from its third identical call on)", train_step,
                  3 * (flops + flops_proj) + flops + flops_proj, True))

    def train_step_r3():
        net_t.project_once_in_training = False
        try:
            train_step()
        finally:
            net_t.project_once_in_training = True
    cases.append(("training step, round 3's schedule (normalised bank shared, projections recomputed inside every checkpoint)", train_step_r3,
                  3 * (flops + flops_proj) + flops + flops_proj, True))
    gstep = [None]
    gins = [mid.clone().requires_grad_(True)] + [x.clone().requires_grad_(True) for x in res]

    def net_fn(m, *r):
        mm, rr = net_t(m, list(r), feats, mask)
        return (mm,) + tuple(rr)

    def train_graphed():
        if gstep[0] is None:
            from mmfs_amd.graphs import GraphedTrainingStep
            for blk in net_t._blocks():
                blk.gradient_checkpointing = False        # (288 GB: a replayed step is bound by its kernels, and recomputation is kernels)
            gstep[0] = GraphedTrainingStep(net_fn, gins, [torch.ones_like(x) for x in gins], list(net_t.parameters()))
            for blk in net_t._blocks():
                blk.gradient_checkpointing = True
        gstep[0](gins, [torch.ones_like(x) for x in gins])
    cases.append(("training step, whole step replayed as ONE HIP graph (no checkpointing)", train_graphed,
                  3 * (flops + flops_proj), True))
    for label, fn, fl, bwd in cases:
        net_t.train(bwd)
        ms = timed(fn, iters=10, warm=3)
        fam, launches = split(fn, iters=3)
        if "HIP-graph replay" in label:
            sample_step()                          # (the mode change above dropped the kept projections: count a step that has them)
            counted = sample_step
        elif "ONE HIP graph" in label:
            def counted():                         # (the replayed step has no recompute pass)
                for blk in net_t._blocks():
                    blk.gradient_checkpointing = False
                try:
                    train_step()
                finally:
                    for blk in net_t._blocks():
                        blk.gradient_checkpointing = True
        else:
            counted = fn
        analytic, fl = fl, executed_gemm_flops(counted)
        obb = sum(op_bytes(B, t, 16, 64, 4 * n, 8, S * n, backward=bwd) for _, t in tokens) + (ob if bwd else 0)
        print(json.dumps({
            "config": "cfg4", "what": f"MMFSNet, 13 blocks at 512 px, B={B}, n_images={n}, bf16, {label}",
            "ms": round(ms, 3), "kernel_us": fam, "launches": launches,
            "gemm_flops": fl, "gemm_flops_reference_schedule": analytic,
            "gemm_mfma_util": round(fl / (fam["gemm"] * 1e-6) / MFMA_PEAK, 4) if fam["gemm"] else None,
            "op_algorithmic_bytes": obb, "op_hbm_frac": round(obb / (fam["op"] * 1e-6) / HBM_PEAK, 4) if fam["op"] else None,
        }), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg3", "cfg4"]
    if "cfg3" in which:
        cfg3()
    if "cfg4" in which:
        cfg4()
