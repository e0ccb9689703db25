"""GPU: the blocks' calls as recorded HIP graphs (mmfs_amd/graphed.py) -- what the unchanged trainer's eager, checkpointed
step runs from its third identical call on.  Held to the REFERENCE's training step (the golden fixture: every input and all
119 parameter gradients), to the eager path call for call, and to the orders of calls a trainer may produce (gradient
accumulation, two forwards before the first backward, a changing shape)."""
import contextlib
import copy
import io
import types

import numpy as np
import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a, dtype):
    t = torch.from_numpy(np.asarray(a))
    return (t.to(dtype) if t.is_floating_point() else t).to(DEV)


def load_params(module, z):
    sd = {k[len("param."):]: torch.from_numpy(np.asarray(v)) for k, v in z.items() if k.startswith("param.")}
    module.load_state_dict(sd, strict=False)
    return module


def rel_err(a, b):
    a = a.detach().double().cpu().numpy(); b = np.asarray(b, np.float64).reshape(a.shape)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def small_net(z, dtype=torch.float32):
    from mmfs_amd.blocks import MMFSNet
    with contextlib.redirect_stdout(io.StringIO()):
        net = load_params(MMFSNet(input_channel=32, block_out_channels=[16, 24], layers_per_block=2,
                                  downsample_factor=8, n_levels=3, n_points=2, gradient_checkpointing=True,
                                  spatial_shapes=[64, 32, 16]), z).to(DEV, dtype)
    return net.train()


def net_step(net, z, dtype=torch.float32, scale=1.0):
    res_t = [(T(z[f"res.{i}"], dtype) * scale).requires_grad_(True) for i in range(6)]
    feats_t = [T(z[f"feat.{i}"], dtype).requires_grad_(True) for i in range(3)]
    mid_t = (T(z["mid"], dtype) * scale).requires_grad_(True)
    new_mid, new_res = net(mid_t, res_t, feats_t, T(z["ms_mask"], None))
    loss = (new_mid.float() * T(z["train.cot_mid"], torch.float32)).sum()
    for i, r in enumerate(new_res):
        loss = loss + (r.float() * T(z[f"train.cot_res.{i}"], torch.float32)).sum()
    loss.backward()
    return new_mid, new_res, mid_t, res_t, feats_t


@pytest.mark.parametrize("keep", [False, True], ids=["recompute", "keep"])
@pytest.mark.parametrize("once", [True, False])
def test_replayed_training_step_matches_the_reference(once, keep):
    """The fixture's reference training step (checkpointing on) through ``MMFSNet`` five times: from the third step on every
    block's forward and its recompute + backward are graph replays -- and the fifth step still returns the reference's output
    and the reference's gradient for every input and each of the 119 parameters."""
    from mmfs_amd import graphed
    z = load_golden("block_sd_mmfs_net")
    net = small_net(z)
    net.project_once_in_training = once
    for blk in net._blocks():                  # the recorded call recomputes like the checkpoint / keeps its activations
        blk.graph_keeps_activations = keep
    before = dict(graphed.stats)
    for step in range(5):
        net.zero_grad(set_to_none=True)
        new_mid, new_res, mid_t, res_t, feats_t = net_step(net, z)
    assert graphed.stats["captures"] - before["captures"] == 7, graphed.stats          # one entry per block
    assert graphed.stats["replays"] - before["replays"] == 7 * 2 * 3                    # steps 3-5: forward + backward each
    assert graphed.stats["eager_backward"] == before["eager_backward"] and graphed.stats["refused"] == before["refused"]
    assert rel_err(new_mid, z["new_mid"]) <= 2e-5 and rel_err(mid_t.grad, z["train.grad_mid"]) <= 1e-4
    for i in range(6):
        assert rel_err(new_res[i], z[f"new_res.{i}"]) <= 2e-5
        assert rel_err(res_t[i].grad, z[f"train.grad_res.{i}"]) <= 1e-4, i
    for i in range(3):
        assert rel_err(feats_t[i].grad, z[f"train.grad_feat.{i}"]) <= 1e-4, i
    want = {k[len("train.grad."):]: v for k, v in z.items() if k.startswith("train.grad.")}
    got = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    assert sorted(got) == sorted(want) and len(want) == 119
    for k, v in want.items():
        assert rel_err(got[k], v) <= 2e-4, f"{k}: {rel_err(got[k], v):.3e}"


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_replays_follow_new_inputs_and_accumulate_gradients(dtype, tol):
    """Steps 4 and 5 run on OTHER inputs than the recorded ones, and without ``zero_grad`` between them (two micro-batches
    of an accumulating trainer): outputs and accumulated gradients against the eager checkpointed path on a twin net."""
    from mmfs_amd import graphed
    z = load_golden("block_sd_mmfs_net")
    net, twin = small_net(z, dtype), small_net(z, dtype)
    for blk in twin._blocks():
        blk.graph_checkpoints = False
    torch.manual_seed(3)
    with torch.no_grad():
        for a, b in zip(net._blocks(), twin._blocks()):
            a.conv.weight.normal_(0, 0.3)
            b.conv.weight.copy_(a.conv.weight)
    before = dict(graphed.stats)
    for step in range(3):
        net.zero_grad(set_to_none=True)
        net_step(net, z, dtype)
    net.zero_grad(set_to_none=True); twin.zero_grad(set_to_none=True)
    outs = {}
    for name, m in (("graphs", net), ("eager", twin)):
        a = net_step(m, z, dtype, scale=0.5)
        b = net_step(m, z, dtype, scale=-1.5)
        outs[name] = ([a[0].detach(), b[0].detach()] + [r.detach() for r in a[1] + b[1]] + [a[2].grad, b[2].grad]
                      + [r.grad for r in a[3] + b[3]] + [f.grad for f in a[4] + b[4]],
                      {k: p.grad for k, p in m.named_parameters() if p.grad is not None})
    assert graphed.stats["replays"] - before["replays"] == 7 * 2 * 3 and graphed.stats["eager_backward"] == before["eager_backward"]
    for x, y in zip(outs["graphs"][0], outs["eager"][0]):
        assert float((x.float() - y.float()).norm() / y.float().norm().clamp_min(1e-6)) <= tol
    assert sorted(outs["graphs"][1]) == sorted(outs["eager"][1]) and len(outs["eager"][1]) == 119
    for k, y in outs["eager"][1].items():
        x = outs["graphs"][1][k]
        assert float((x.float() - y.float()).norm() / y.float().norm().clamp_min(1e-6)) <= tol, k


@pytest.mark.parametrize("keep", [False, True], ids=["recompute", "keep"])
def test_two_forwards_before_the_first_backward(keep):
    """A block called twice before either backward runs (two losses on one net): the second forward overwrites the first
    call's static inputs.  A recorded call that recomputes copies the caller's tensors in again and replays; one that keeps
    its activations has lost the first call's and recomputes that backward eagerly."""
    from mmfs_amd import graphed
    from mmfs_amd.blocks import MMFSBlock
    z = load_golden("block_sd_mmfs_block")
    with contextlib.redirect_stdout(io.StringIO()):
        blk = load_params(MMFSBlock(attn_dim=32, query_dim=16, feat_dim=32, num_heads=4, n_points=2, n_levels=3,
                                    gradient_checkpointing=True, grid_size=8, spatial_shapes=[8, 4, 2],
                                    base_spatial_shape=4, max_num_image_per_seq=5), z).to(DEV).train()
    with torch.no_grad():
        blk.conv.weight.normal_(0, 0.3)
    blk.graph_keeps_activations = keep
    twin = copy.deepcopy(blk)
    twin.graph_checkpoints = False
    shapes, mask, go = [(8, 8), (4, 4), (2, 2)], T(z["ms_mask"], None), T(z["grad_out"], torch.float32)

    def pair(m):
        m.zero_grad(set_to_none=True)
        s1 = T(z["sample"], torch.float32).requires_grad_(True)
        s2 = (T(z["sample"], torch.float32) * -0.7).requires_grad_(True)
        f = T(z["ms_feat"], torch.float32).requires_grad_(True)
        o1 = m(s1, f, mask, shapes)
        o2 = m(s2, f, mask, shapes)
        o1.backward(go)
        o2.backward(go * 2.0)
        return [o1.detach(), o2.detach(), s1.grad, s2.grad, f.grad] + [p.grad for p in m.parameters() if p.grad is not None]

    before = dict(graphed.stats)
    for _ in range(3):
        got = pair(blk)
    want = pair(twin)
    assert graphed.stats["captures"] - before["captures"] == 1 and graphed.stats["replays"] > before["replays"]
    assert graphed.stats["eager_backward"] - before["eager_backward"] == (2 if keep else 0)      # (the first call of pairs 2 and 3)
    assert len(got) == len(want)
    for x, y in zip(got, want):
        assert float((x - y).abs().max()) <= 1e-5 * max(1.0, float(y.abs().max()))


def llama_layer(dtype=torch.float32):
    from mmfs_amd.blocks import LlamaMMFSAttention
    cfg = types.SimpleNamespace(hidden_size=512, num_attention_heads=8, rms_norm_eps=1e-6, max_position_embeddings=64,
                                image_embed_dim=128, spatial_shapes=[8, 4, 2])
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        layer = LlamaMMFSAttention(cfg, 0)
    with torch.no_grad():
        layer.gate.fill_(0.7)
        layer.attn.sampling_offsets.weight.normal_(0, 0.02)
        layer.attn.attention_weights.weight.normal_(0, 0.02)
    return layer.to(DEV, dtype).train()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_llama_layer_training_calls_replay(dtype, tol):
    """``LlamaMMFSAttention`` in training mode (``recompute=False``: the forward graph keeps its activations, the backward
    graph is the backward alone): steps 4-5 on new inputs, accumulating, against the eager layer; then two forwards before
    the first backward -- the first call's activations are gone, its backward recomputes eagerly -- and a no-grad call
    (the first pass of a checkpointing decoder layer)."""
    from mmfs_amd import graphed
    layer = llama_layer(dtype)
    twin = copy.deepcopy(layer)
    twin.graph_training_calls = False
    B, Lq, n, hw = 2, 33, 2, 64 + 16 + 4
    g = torch.Generator().manual_seed(1)
    hidden = torch.randn(B, Lq, 512, generator=g).to(DEV, dtype)
    feats = torch.randn(B, n, hw, 128, generator=g).to(DEV, dtype)
    mask = torch.ones(B, Lq, n, device=DEV, dtype=dtype)
    mask[0, :10, 1] = 0.0
    go = torch.randn(B, Lq, 512, generator=g).to(DEV, dtype)

    def step(m, scale, zero=True):
        if zero:
            m.zero_grad(set_to_none=True)
        x = (hidden * scale).requires_grad_(True)
        f = feats.clone().requires_grad_(True)
        y = m(x, f, mask, residual=x)
        y.backward(go)
        return [y.detach(), x.grad, f.grad]

    def close(a, b):
        for x, y in zip(a, b):
            assert float((x.float() - y.float()).norm() / y.float().norm().clamp_min(1e-6)) <= tol

    before = dict(graphed.stats)
    for _ in range(3):
        step(layer, 1.0)
    assert graphed.stats["captures"] - before["captures"] == 1, graphed.stats
    got = step(layer, 0.5) + step(layer, -1.5, zero=False) + [p.grad for p in layer.parameters() if p.grad is not None]
    want = step(twin, 0.5) + step(twin, -1.5, zero=False) + [p.grad for p in twin.parameters() if p.grad is not None]
    assert graphed.stats["replays"] - before["replays"] == 6 and graphed.stats["eager_backward"] == before["eager_backward"]
    assert len(got) == len(want) >= 6 + 13
    close(got, want)

    # two forwards, then the backwards in the order of the forwards
    def pair(m):
        m.zero_grad(set_to_none=True)
        x1 = (hidden * 0.3).requires_grad_(True); x2 = (hidden * 1.1).requires_grad_(True)
        y1 = m(x1, feats, mask, residual=x1); y2 = m(x2, feats, mask, residual=x2)
        y1.backward(go); y2.backward(go)
        return [y1.detach(), y2.detach(), x1.grad, x2.grad] + [p.grad for p in m.parameters() if p.grad is not None]
    for _ in range(3):                       # (feats without a gradient: another key, recorded on its third call)
        got = pair(layer)
    assert graphed.stats["eager_backward"] - before["eager_backward"] == 2, graphed.stats      # (the first call of pairs 2 and 3)
    close(got, pair(twin))

    with torch.no_grad():
        for _ in range(4):
            y = layer(hidden, feats, mask, residual=hidden)
        close([y], [twin(hidden, feats, mask, residual=hidden)])
    assert graphed.stats["refused"] == before["refused"]


def test_a_changed_switch_is_another_key():
    """What a recorded call depends on beside its tensors is part of its key: a block whose path switch is flipped does not
    replay the old graphs; ``graphed.enabled = False`` turns the mechanism off."""
    from mmfs_amd import graphed
    from mmfs_amd.blocks import MMFSBlock
    z = load_golden("block_sd_mmfs_block")
    with contextlib.redirect_stdout(io.StringIO()):
        blk = load_params(MMFSBlock(attn_dim=32, query_dim=16, feat_dim=32, num_heads=4, n_points=2, n_levels=3,
                                    gradient_checkpointing=True, grid_size=8, spatial_shapes=[8, 4, 2],
                                    base_spatial_shape=4, max_num_image_per_seq=5), z).to(DEV).train()
    shapes, mask = [(8, 8), (4, 4), (2, 2)], T(z["ms_mask"], None)
    s = T(z["sample"], torch.float32)
    f = T(z["ms_feat"], torch.float32)

    def call(sample):
        x = sample.clone().requires_grad_(True)
        out = blk(x, f, mask, shapes)
        out.sum().backward()
        return out
    before = dict(graphed.stats)
    for _ in range(4):
        call(s)
    assert graphed.stats["captures"] - before["captures"] == 1
    r = graphed.stats["replays"]
    blk.mmfs.stack_heads_in_training = False
    call(s)
    assert graphed.stats["replays"] == r                       # (eager: a new key, counting)
    blk.mmfs.stack_heads_in_training = True
    call(s)
    assert graphed.stats["replays"] == r + 2
    graphed.enabled = False
    try:
        call(s)
        assert graphed.stats["replays"] == r + 2
    finally:
        graphed.enabled = True
    # a mode change moves the package's cache epoch: what was recorded under the old one is dropped with its buffers,
    # the calls count again and are recorded again
    blk.eval(); blk.train()
    call(s)
    assert graphed.stats["replays"] == r + 2 and len(blk.__dict__["_graphed"]) == 1
    call(s); call(s)
    assert graphed.stats["captures"] - before["captures"] == 2 and graphed.stats["replays"] == r + 4


@pytest.mark.parametrize("reentrant", [True, False])
def test_llama_layer_inside_the_decoders_checkpoint(reentrant):
    """What the reference's decoder does with every layer in training (modeling_llama_mmfs.py:700-717:
    ``torch.utils.checkpoint.checkpoint(custom_forward, hidden, vision, mask, ...)``, the reentrant flavour by default): the
    MMFS layer is first called WITHOUT autograd, then again with it from inside the backward pass -- where its graphs are
    also recorded, on the autograd engine's thread.  Five steps against the same stack with the graphs off."""
    import torch.utils.checkpoint as cp
    from mmfs_amd import graphed
    layer = llama_layer(torch.float32)
    twin = copy.deepcopy(layer)
    twin.graph_training_calls = False
    B, Lq, n, hw = 2, 33, 2, 64 + 16 + 4
    g = torch.Generator().manual_seed(2)
    hidden = torch.randn(B, Lq, 512, generator=g).to(DEV)
    feats = torch.randn(B, n, hw, 128, generator=g).to(DEV)
    mask = torch.ones(B, Lq, n, device=DEV)
    go = torch.randn(B, Lq, 512, generator=g).to(DEV)
    dense = torch.nn.Linear(512, 512).to(DEV)              # (a stand-in for the decoder layer's own sub-layers)

    def step(m, scale):
        m.zero_grad(set_to_none=True)
        dense.zero_grad(set_to_none=True)
        x = (hidden * scale).requires_grad_(True)
        f = feats.clone().requires_grad_(True)

        def decoder_layer(h, v, c):
            h = h + dense(h)
            return m(h, v, c, residual=h)
        y = cp.checkpoint(decoder_layer, x, f, mask, use_reentrant=reentrant)
        y.backward(go)
        return [y.detach(), x.grad, f.grad, dense.weight.grad.clone()] + [p.grad for p in m.parameters() if p.grad is not None]

    before = dict(graphed.stats)
    for i in range(5):
        got = step(layer, 0.5 + 0.25 * i)
    want = step(twin, 1.5)
    if reentrant:
        assert graphed.stats["captures"] > before["captures"] and graphed.stats["replays"] > before["replays"], graphed.stats
    else:
        # a non-reentrant checkpoint around the layer matches the tensors saved by its forward and by its recomputation
        # one by one: the layer stays on the plain path in both
        assert graphed.stats["replays"] == before["replays"]
    assert graphed.stats["refused"] == before["refused"]
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))


def test_a_device_bound_call_is_not_replayed():
    """``graphed_call`` knows the host time of the plain path (it runs it) and times the recorded forward on an idle device:
    a call whose kernels outlast their launches gives its graphs back and stays on the plain path (``device_bound_ratio``;
    forced here -- at these sizes every call is bound by the host)."""
    from mmfs_amd import graphed
    layer = llama_layer(torch.float32)
    hidden = torch.randn(2, 33, 512, device=DEV)
    feats = torch.randn(2, 2, 84, 128, device=DEV)
    mask = torch.ones(2, 33, 2, device=DEV)
    before = dict(graphed.stats)
    graphed.device_bound_ratio = 0.0
    try:
        for _ in range(5):
            x = hidden.clone().requires_grad_(True)
            layer(x, feats, mask, residual=x).sum().backward()
    finally:
        graphed.device_bound_ratio = 1.0
    assert graphed.stats["device_bound"] - before["device_bound"] == 1
    assert graphed.stats["captures"] == before["captures"] and graphed.stats["replays"] == before["replays"]
    assert graphed.stats["refused"] == before["refused"]
    entry, = layer.__dict__["_graphed"].values()
    assert entry.state == -2 and not hasattr(entry, "fwd")


def test_optimizer_steps_between_calls_do_not_re_record(monkeypatch):
    """the key of a recorded call held the VERSION of trainable parameters whenever gradients were disabled -- the
    no-grad forward of a reentrant-checkpointed decoder layer -- and every ``optimizer.step()`` bumps it: each layer re-recorded
    every step.  Trainable parameters are in the key by address only: with an optimizer step between the steps the captures
    stay flat and the replays continue (and the results follow the moved weights: the graphs read them where they are)."""
    import torch.utils.checkpoint as cp
    from mmfs_amd import graphed
    layer = llama_layer(torch.float32)
    twin = copy.deepcopy(layer)
    twin.graph_training_calls = False
    B, Lq, n, hw = 2, 33, 2, 64 + 16 + 4
    g = torch.Generator().manual_seed(4)
    hidden = torch.randn(B, Lq, 512, generator=g).to(DEV)
    feats = torch.randn(B, n, hw, 128, generator=g).to(DEV)
    mask = torch.ones(B, Lq, n, device=DEV)
    go = torch.randn(B, Lq, 512, generator=g).to(DEV)
    opts = [torch.optim.SGD(m.parameters(), lr=1e-3) for m in (layer, twin)]

    def step(m, opt):
        opt.zero_grad(set_to_none=True)
        x = hidden.clone().requires_grad_(True)
        f = feats.clone().requires_grad_(True)
        y = cp.checkpoint(lambda h, v, c: m(h, v, c, residual=h), x, f, mask, use_reentrant=True)
        y.backward(go)
        opt.step()
        return y.detach()

    for _ in range(4):                                     # warm-up + recording
        step(layer, opts[0]); step(twin, opts[1])
    before = dict(graphed.stats)
    for _ in range(6):
        got = step(layer, opts[0]); want = step(twin, opts[1])
    assert graphed.stats["captures"] == before["captures"], (before, graphed.stats)
    assert graphed.stats["replays"] >= before["replays"] + 6, (before, graphed.stats)
    assert graphed.stats["refused"] == before["refused"]
    assert float((got - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))

