"""Module-level parity on CPU: the product's MMFS / blocks / bank builders (host logic)
against golden vectors captured from the reference (tests/golden/make_golden.py), with the
CPU oracle standing in for the HIP op (tests may use the oracle; the product never does).
The same checks run on the GPU with the real op in test_modules_gpu.py."""
import ast
import contextlib
import io
import types

import numpy as np
import pytest
import torch

from helpers import load_golden
from oracle.msda_oracle import OracleMSDAFunction


@pytest.fixture()
def oracle_op(monkeypatch):
    import mmfs_amd.modules.mmfs as m1
    import mmfs_amd.modules.ms_deform_attn as m2
    monkeypatch.setattr(m1, "MSDeformAttnFunction", OracleMSDAFunction)
    monkeypatch.setattr(m2, "MSDeformAttnFunction", OracleMSDAFunction)


def T(a, dtype=torch.float64):
    t = torch.from_numpy(np.asarray(a))
    return t.to(dtype) if t.is_floating_point() else t


def load_params(module, z, dtype=torch.float64):
    sd = {k[len("param."):]: T(v, dtype) for k, v in z.items() if k.startswith("param.")}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected                       # reference keys all exist here
    assert all(k.endswith("scale_ratios") for k in missing), missing
    return module


def close(a, b, tol):
    a = a.detach().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    err = np.abs(a - b.reshape(a.shape)).max() if a.size else 0.0
    assert err <= tol * max(1.0, np.abs(b).max()), f"max err {err:.3e}"


MMFS_CASES = ["mmfs_llm_mask3d", "mmfs_llm_decode", "mmfs_llm_n1", "mmfs_sd_mask2d",
              # the point counts the decoders really use (P = 8) and the north star's (P = 4)
              "mmfs_p8_llm_n3", "mmfs_p8_llm_n4", "mmfs_p8_llm_n1", "mmfs_p8_llm_decode", "mmfs_p4_llm_n3",
              "mmfs_p8_sd_grid", "mmfs_p8_sd_n1"]


def build_mmfs(z, dtype=torch.float64):
    from mmfs_amd.modules import MMFS
    cfg = ast.literal_eval(str(z["cfg"]))
    with contextlib.redirect_stdout(io.StringIO()):
        m = MMFS(**cfg).to(dtype)
    return load_params(m, z, dtype), cfg


@pytest.mark.parametrize("name", MMFS_CASES)
def test_mmfs_forward_backward_matches_reference(name, oracle_op):
    z = load_golden(name)
    m, cfg = build_mmfs(z)
    q = T(z["query"]).requires_grad_(True)
    f = T(z["feat"]).requires_grad_(True)
    out = m(q, T(z["reference_points"]), f, T(z["spatial_shapes"]), T(z["level_start_index"]), None,
            T(z["attention_mask"], torch.float32) if z["attention_mask"].dtype.kind == "f" else T(z["attention_mask"]))
    close(out, z["out"], 1e-10)
    out.backward(T(z["grad_out"]))
    close(q.grad, z["grad_query"], 1e-9)
    close(f.grad, z["grad_feat"], 1e-9)
    for k, p in m.named_parameters():
        if "grad." + k in z:
            close(p.grad, z["grad." + k], 1e-9)
        else:
            assert p.grad is None or not p.grad.any(), k   # e.g. frozen ignore_token


@pytest.mark.parametrize("name", ["mmfs_p8_llm_boxes", "mmfs_p8_sd_padded"])
def test_mmfs_box_reference_points_and_padding_mask_match_reference(name, oracle_op):
    """The two API-fidelity branches of MMFS.forward that no caller of the reference takes: 4-D box reference
    points (ops/modules/mmfs.py:251-258) and ``input_padding_mask`` (mmfs.py:165-172), against the reference's outputs
    and every gradient (VERDICT r5 missing 3: only the encoder twin had such a golden)."""
    z = load_golden(name)
    m, cfg = build_mmfs(z)
    q = T(z["query"]).requires_grad_(True)
    f = T(z["feat"]).requires_grad_(True)
    pad = torch.from_numpy(z["input_padding_mask"]) if "input_padding_mask" in z else None
    mask = T(z["attention_mask"], torch.float32) if z["attention_mask"].dtype.kind == "f" else T(z["attention_mask"])
    out = m(q, T(z["reference_points"]), f, T(z["spatial_shapes"]), T(z["level_start_index"]), pad, mask)
    close(out, z["out"], 1e-10)
    out.backward(T(z["grad_out"]))
    close(q.grad, z["grad_query"], 1e-9)
    close(f.grad, z["grad_feat"], 1e-9)
    for k, p in m.named_parameters():
        if "grad." + k in z:
            close(p.grad, z["grad." + k], 1e-9)
    if pad is not None:                                   # a padded token receives no gradient through the value projection ...
        assert float(f.grad[pad].abs().max()) == 0.0
    assert z["reference_points"].shape[-1] == (4 if "boxes" in name else 2)


def test_mmfs_fp32_within_1e5(oracle_op):
    z = load_golden("mmfs_llm_mask3d_f32")
    m, _ = build_mmfs(z, torch.float32)
    out = m(T(z["query"], torch.float32), T(z["reference_points"], torch.float32), T(z["feat"], torch.float32),
            T(z["spatial_shapes"]), T(z["level_start_index"]), None, T(z["attention_mask"], torch.float32))
    close(out, load_golden("mmfs_llm_mask3d")["out"], 1e-5)    # vs the fp64 reference run


def test_mmfs_state_dict_keys_are_the_references():
    from mmfs_amd.modules import MMFS
    with contextlib.redirect_stdout(io.StringIO()):
        m = MMFS(d_model=32, d_query=16, d_value=8, d_out=16, n_levels=2, n_heads=4, n_points=2,
                 spatial_shapes=[4, 2], base_spatial_shape=4, max_num_image_per_seq=5)
    assert sorted(m.state_dict()) == sorted([
        "sampling_offsets.weight", "sampling_offsets.bias", "ignore_token",
        "dynamic_offset_mask.weight", "dynamic_offset_mask.bias", "attention_weights.weight",
        "attention_weights.bias", "value_proj.weight", "value_proj.bias", "output_proj.weight",
        "output_proj.bias", "query_relpos.weight"])
    assert not m.ignore_token.requires_grad and m.sampling_offsets.weight.abs().sum() == 0
    assert m.sampling_offsets.bias.abs().max() <= 3 and m.im2col_step == 1


def test_mmfs_known_answers(oracle_op):
    """all images masked -> output == output_proj.bias; masked images get zero
    attention; attention over real points sums to < 1 (the sinks keep the rest)."""
    z = load_golden("mmfs_sd_mask2d")
    m, cfg = build_mmfs(z)
    q, f = T(z["query"]), T(z["feat"])
    mask = torch.zeros(q.shape[0], f.shape[1], dtype=torch.long)
    with torch.no_grad():
        m.ignore_token.zero_()
        out = m(q, T(z["reference_points"]), f, T(z["spatial_shapes"]), T(z["level_start_index"]), None, mask)
        assert torch.allclose(out, m.output_proj.bias.expand_as(out), atol=1e-12)
        loc, attn, sink = m.sampling_plan(q, T(z["reference_points"]), T(z["spatial_shapes"]),
                                          T(z["attention_mask"]), f.shape[1])
    L = cfg["n_levels"]
    am = T(z["attention_mask"]).bool()                    # [B, n]
    per_image = attn.reshape(*attn.shape[:3], f.shape[1], L, -1).sum((-1, -2))   # [B, Lq, H, n]
    assert (per_image[~am[:, None, None, :].expand_as(per_image)] < 1e-300).all()
    total = attn.sum((-1, -2)) + sink
    assert torch.allclose(total, torch.ones_like(total)) and (attn.sum((-1, -2)) < 1).all()
    # the normalised location is the same at every level of one image (scale_ratios cancel)
    loc5 = loc.reshape(*loc.shape[:3], f.shape[1], L, *loc.shape[4:])
    assert torch.allclose(loc5, loc5[:, :, :, :, :1].expand_as(loc5), atol=1e-12)


def test_llama_mmfs_attention_matches_reference(oracle_op):
    from mmfs_amd.blocks import LlamaMMFSAttention
    z = load_golden("block_llama_mmfs_attention")
    cfg = types.SimpleNamespace(hidden_size=64, num_attention_heads=4, rms_norm_eps=1e-6,
                                max_position_embeddings=64, image_embed_dim=32, spatial_shapes=[8, 4, 2])
    with contextlib.redirect_stdout(io.StringIO()):
        att = load_params(LlamaMMFSAttention(cfg, layer_idx=0).double(), z)
    assert sorted(k.split(".")[0] for k in att.state_dict()) .count("attn") == 12
    h = T(z["hidden"]).requires_grad_(True)
    f = T(z["feats"]).requires_grad_(True)
    out = att(h, f, T(z["mask"], torch.float32))
    close(out, z["out"], 1e-10)
    out.backward(T(z["grad_out"]))
    close(h.grad, z["grad_hidden"], 1e-9)
    close(f.grad, z["grad_feats"], 1e-9)
    for k, p in att.named_parameters():
        if "grad." + k in z:
            close(p.grad, z["grad." + k], 1e-9)


def test_sd_mmfs_block_matches_reference(oracle_op):
    from mmfs_amd.blocks import MMFSBlock
    z = load_golden("block_sd_mmfs_block")
    with contextlib.redirect_stdout(io.StringIO()):
        blk = MMFSBlock(attn_dim=32, query_dim=16, feat_dim=32, num_heads=4, n_points=2, n_levels=3,
                        gradient_checkpointing=False, grid_size=8, spatial_shapes=[8, 4, 2],
                        base_spatial_shape=4, max_num_image_per_seq=5).double()
    # the sin-cos table built here equals the reference's (stored in the fixture)
    close(blk.pos_embed, z["param.pos_embed"], 1e-6)
    load_params(blk, z)
    s = T(z["sample"]).requires_grad_(True)
    f = T(z["ms_feat"]).requires_grad_(True)
    out = blk(s, f, T(z["ms_mask"]), [(8, 8), (4, 4), (2, 2)])
    close(out, z["out"], 1e-10)
    out.backward(T(z["grad_out"]))
    close(s.grad, z["grad_sample"], 1e-9)
    close(f.grad, z["grad_ms_feat"], 1e-9)
    for k, p in blk.named_parameters():
        if "grad." + k in z:
            close(p.grad, z["grad." + k], 1e-9)


def test_sd_mmfs_net_matches_reference(oracle_op):
    from mmfs_amd.blocks import MMFSNet
    z = load_golden("block_sd_mmfs_net")
    with contextlib.redirect_stdout(io.StringIO()):
        net = MMFSNet(input_channel=32, block_out_channels=[16, 24], layers_per_block=2,
                      downsample_factor=8, n_levels=3, n_points=2, gradient_checkpointing=False,
                      spatial_shapes=[64, 32, 16]).double()
    assert len(net.mmfs_down_blocks) == 6
    ref_keys = sorted(k[len("param."):] for k in z if k.startswith("param."))
    assert sorted(net.state_dict()) == ref_keys               # checkpoint compatible
    load_params(net, z)
    res = [T(z[f"res.{i}"]) for i in range(6)]
    feats = [T(z[f"feat.{i}"]) for i in range(3)]
    with torch.no_grad():
        mid, new_res = net(T(z["mid"]), res, feats, T(z["ms_mask"]))
    close(mid, z["new_mid"], 1e-10)
    for i, r in enumerate(new_res):
        close(r, z[f"new_res.{i}"], 1e-10)


def _tiny_net(z, **kw):
    from mmfs_amd.blocks import MMFSNet
    with contextlib.redirect_stdout(io.StringIO()):
        net = MMFSNet(input_channel=32, block_out_channels=[16, 24], layers_per_block=2,
                      downsample_factor=8, n_levels=3, n_points=2, gradient_checkpointing=False,
                      spatial_shapes=[64, 32, 16], **kw).double()
    load_params(net, z)
    torch.manual_seed(3)
    with torch.no_grad():                         # the fixture's conv is the reference's zero init
        for blk in net._blocks():
            blk.conv.weight.normal_(0, 0.3)
    return net


def test_sd_mmfs_net_fused_schedule_equals_the_references(oracle_op):
    """One normalisation of the bank + folded affines (MMFSNet.fused_schedule) against the
    reference's schedule (13 LayerNorms + 13 projections): outputs, input grads, every parameter grad."""
    z = load_golden("block_sd_mmfs_net")
    outs = []
    for fused in (False, True):
        net = _tiny_net(z)
        net.fused_schedule = fused
        res = [T(z[f"res.{i}"]).requires_grad_(True) for i in range(6)]
        feats = [T(z[f"feat.{i}"]).requires_grad_(True) for i in range(3)]
        mid = T(z["mid"]).requires_grad_(True)
        new_mid, new_res = net(mid, res, feats, T(z["ms_mask"]))
        g = torch.Generator().manual_seed(5)
        loss = (new_mid * torch.randn(new_mid.shape, generator=g, dtype=torch.float64)).sum()
        for r in new_res:
            loss = loss + (r * torch.randn(r.shape, generator=g, dtype=torch.float64)).sum()
        loss.backward()
        outs.append(dict(mid=new_mid.detach(), res=[r.detach() for r in new_res],
                         gfeat=[f.grad for f in feats], gres=[r.grad for r in res],
                         gparam={k: p.grad for k, p in net.named_parameters() if p.grad is not None}))
    a, b = outs
    close(b["mid"], a["mid"], 1e-11)
    for x, y in zip(b["res"] + b["gfeat"] + b["gres"], a["res"] + a["gfeat"] + a["gres"]):
        close(x, y, 1e-10)
    assert sorted(a["gparam"]) == sorted(b["gparam"])
    assert any("feat_norm.weight" in k for k in a["gparam"])
    for k in a["gparam"]:
        close(b["gparam"][k], a["gparam"][k], 1e-9)


def _net_training_step(net, z, dtype=torch.float64, dev="cpu"):
    """One training step of the tiny net on the fixture's inputs and cotangents -> outputs, input grads, parameter grads."""
    conv = lambda a: T(a, dtype).to(dev) if np.asarray(a).dtype.kind == "f" else T(a).to(dev)
    res = [conv(z[f"res.{i}"]).requires_grad_(True) for i in range(6)]
    feats = [conv(z[f"feat.{i}"]).requires_grad_(True) for i in range(3)]
    mid = conv(z["mid"]).requires_grad_(True)
    new_mid, new_res = net(mid, res, feats, conv(z["ms_mask"]))
    loss = (new_mid * conv(z["train.cot_mid"])).sum()
    for i, r in enumerate(new_res):
        loss = loss + (r * conv(z[f"train.cot_res.{i}"])).sum()
    loss.backward()
    return dict(mid=new_mid.detach(), res=[r.detach() for r in new_res], gmid=mid.grad, gres=[r.grad for r in res],
                gfeat=[f.grad for f in feats], gparam={k: p.grad for k, p in net.named_parameters() if p.grad is not None})


@pytest.mark.parametrize("schedule", ["reference", "shared_bank", "project_once"])
def test_sd_mmfs_net_training_step_matches_the_reference_gradients(oracle_op, schedule):
    """the training ASSEMBLY against the reference itself, not against last round's schedule.  The
    fixture carries a training step of the reference's MMFSNet (gradient checkpointing on in every block,
    sd_mmfs.py:138-141): cotangents for the seven outputs, the gradient of every input and of all 119 parameters.  Each of
    the build's training schedules -- the reference's own (LayerNorm + projection inside every checkpoint), the shared
    un-affined normalisation, the bank projected once for all blocks outside the checkpoints (the default where memory
    allows) -- reproduces them in fp64."""
    from mmfs_amd.blocks import MMFSNet
    z = load_golden("block_sd_mmfs_net")
    with contextlib.redirect_stdout(io.StringIO()):
        net = MMFSNet(input_channel=32, block_out_channels=[16, 24], layers_per_block=2, downsample_factor=8, n_levels=3,
                      n_points=2, gradient_checkpointing=True, spatial_shapes=[64, 32, 16]).double()
    load_params(net, z)
    net.train()
    net.share_normalised_bank = schedule != "reference"
    net.fused_schedule = schedule != "reference"
    net.project_once_in_training = schedule == "project_once"
    got = _net_training_step(net, z)
    close(got["mid"], z["new_mid"], 1e-11)
    for i in range(6):
        close(got["res"][i], z[f"new_res.{i}"], 1e-11)
        close(got["gres"][i], z[f"train.grad_res.{i}"], 1e-10)
    close(got["gmid"], z["train.grad_mid"], 1e-10)
    for i in range(3):
        close(got["gfeat"][i], z[f"train.grad_feat.{i}"], 1e-9)
    want = {k[len("train.grad."):]: v for k, v in z.items() if k.startswith("train.grad.")}
    assert sorted(got["gparam"]) == sorted(want) and len(want) == 119
    for k, v in want.items():
        close(got["gparam"][k], v, 1e-9)


def test_sd_mmfs_net_shares_the_normalised_bank_under_checkpointing(oracle_op):
    """Training with gradient checkpointing: the bank is normalised once without the affine and every block folds
    its own affine into the projection it recomputes inside its checkpoint (MMFSNet.share_normalised_bank) --
    against the reference's schedule (LayerNorm + projection inside every checkpoint): outputs, input gradients,
    every parameter gradient; and the projected banks are not among the tensors the step keeps."""
    z = load_golden("block_sd_mmfs_net")
    outs = []
    for share in (False, True):
        net = _tiny_net(z)
        net.share_normalised_bank = share
        net.project_once_in_training = False              # (round 3's schedule: everything recomputed inside the checkpoints)
        for blk in net._blocks():
            blk.gradient_checkpointing = True
        net.train()
        seen = []
        hooks = [blk.mmfs.register_forward_pre_hook(lambda m, args, kwargs: seen.append(kwargs.get("value") is not None),
                                                    with_kwargs=True) for blk in net._blocks()]
        res = [T(z[f"res.{i}"]).requires_grad_(True) for i in range(6)]
        feats = [T(z[f"feat.{i}"]).requires_grad_(True) for i in range(3)]
        mid = T(z["mid"]).requires_grad_(True)
        new_mid, new_res = net(mid, res, feats, T(z["ms_mask"]))
        g = torch.Generator().manual_seed(5)
        loss = (new_mid * torch.randn(new_mid.shape, generator=g, dtype=torch.float64)).sum()
        for r in new_res:
            loss = loss + (r * torch.randn(r.shape, generator=g, dtype=torch.float64)).sum()
        loss.backward()
        for h in hooks:
            h.remove()
        assert len(seen) == 14 and all(v == share for v in seen)      # 7 blocks, forward + recompute; folded projection or not
        outs.append(dict(mid=new_mid.detach(), res=[r.detach() for r in new_res],
                         gfeat=[f.grad for f in feats], gres=[r.grad for r in res],
                         gparam={k: p.grad for k, p in net.named_parameters() if p.grad is not None}))
    a, b = outs
    close(b["mid"], a["mid"], 1e-11)
    for x, y in zip(b["res"] + b["gfeat"] + b["gres"], a["res"] + a["gfeat"] + a["gres"]):
        close(x, y, 1e-10)
    assert sorted(a["gparam"]) == sorted(b["gparam"])
    assert any("feat_norm.weight" in k for k in a["gparam"]) and any("feat_norm.bias" in k for k in a["gparam"])
    for k in a["gparam"]:
        close(b["gparam"][k], a["gparam"][k], 1e-9)


def test_sd_mmfs_net_projects_once_in_training(oracle_op):
    """Round 4: a training step under gradient checkpointing projects the bank for all blocks ONCE, as
    one batched GEMM outside the checkpoints (MMFSNet.project_once_in_training, default) -- every block gets its
    projection as a checkpoint input, in the forward and in the recompute pass -- against the reference's schedule
    (LayerNorm + projection inside every checkpoint): outputs, input gradients, EVERY parameter gradient."""
    z = load_golden("block_sd_mmfs_net")
    outs = []
    for once in (False, True):
        net = _tiny_net(z)
        net.project_once_in_training = once
        net.share_normalised_bank = False                 # (the reference's schedule on the other side)
        net.fused_schedule = once
        for blk in net._blocks():
            blk.gradient_checkpointing = True
        net.train()
        seen = []
        hooks = [blk.mmfs.register_forward_pre_hook(lambda m, args, kwargs: seen.append(kwargs.get("value") is not None),
                                                    with_kwargs=True) for blk in net._blocks()]
        res = [T(z[f"res.{i}"]).requires_grad_(True) for i in range(6)]
        feats = [T(z[f"feat.{i}"]).requires_grad_(True) for i in range(3)]
        mid = T(z["mid"]).requires_grad_(True)
        new_mid, new_res = net(mid, res, feats, T(z["ms_mask"]))
        g = torch.Generator().manual_seed(5)
        loss = (new_mid * torch.randn(new_mid.shape, generator=g, dtype=torch.float64)).sum()
        for r in new_res:
            loss = loss + (r * torch.randn(r.shape, generator=g, dtype=torch.float64)).sum()
        loss.backward()
        for h in hooks:
            h.remove()
        assert len(seen) == 14 and all(v == once for v in seen)       # 7 blocks, forward + recompute: handed a projection or not
        outs.append(dict(mid=new_mid.detach(), res=[r.detach() for r in new_res],
                         gfeat=[f.grad for f in feats], gres=[r.grad for r in res],
                         gparam={k: p.grad for k, p in net.named_parameters() if p.grad is not None}))
    a, b = outs
    close(b["mid"], a["mid"], 1e-11)
    for x, y in zip(b["res"] + b["gfeat"] + b["gres"], a["res"] + a["gfeat"] + a["gres"]):
        close(x, y, 1e-10)
    assert sorted(a["gparam"]) == sorted(b["gparam"])
    assert any("feat_norm.weight" in k for k in a["gparam"]) and any("value_proj.bias" in k for k in a["gparam"])
    for k in a["gparam"]:
        close(b["gparam"][k], a["gparam"][k], 1e-9)


def test_sd_mmfs_net_reuses_projections_while_sampling(oracle_op):
    """Outside autograd the projected bank is kept for as long as the caller passes the same,
    unmodified feature tensors and the parameters do not move (denoising loop); anything else
    recomputes.  ProjectedFeatures can also be made once and handed in."""
    z = load_golden("block_sd_mmfs_net")
    net = _tiny_net(z).eval()
    calls = []
    inner = net.project_features
    net.project_features = lambda feats: (calls.append(1), inner(feats))[1]
    res = [T(z[f"res.{i}"]) for i in range(6)]
    feats = [T(z[f"feat.{i}"]) for i in range(3)]
    mask, mid = T(z["ms_mask"]), T(z["mid"])
    with torch.no_grad():
        first = net(mid, res, feats, mask)
        again = net(mid * 0.5, [r * 2 for r in res], feats, mask)       # next denoising step: new samples
        assert len(calls) == 1
        close(net(mid, res, feats, mask)[0], first[0], 0.0)
        assert len(calls) == 1
        feats[1].add_(torch.randn(feats[1].shape, dtype=torch.float64))    # features changed in place
        changed = net(mid, res, feats, mask)
        assert len(calls) == 2 and float((changed[0] - first[0]).abs().max()) > 1e-6
        net.mmfs_mid_block.mmfs.value_proj.weight.mul_(0.9)             # parameters moved
        net(mid, res, feats, mask)
        assert len(calls) == 3
        net(mid, res, [f.clone() for f in feats], mask)                 # other tensors, same content
        assert len(calls) == 4
        proj = inner(feats)                                             # explicit hand-over
        close(net(mid, res, proj, mask)[0], net(mid, res, feats, mask)[0], 0.0)
    net(mid, res, feats, mask)                                          # under autograd: no cache kept
    assert "_projected" not in net.__dict__
    del again


def test_ms_deform_attn_module(oracle_op):
    """Encoder twin: shapes, init (directional bias, zero weights) and agreement with a
    by-hand evaluation through the oracle."""
    from mmfs_amd.modules import MSDeformAttn
    torch.manual_seed(0)
    m = MSDeformAttn(d_model=32, n_levels=2, n_heads=4, n_points=2, ratio=0.5).double()
    assert sorted(m.state_dict()) == sorted([
        "sampling_offsets.weight", "sampling_offsets.bias", "attention_weights.weight",
        "attention_weights.bias", "value_proj.weight", "value_proj.bias", "output_proj.weight", "output_proj.bias"])
    b = m.sampling_offsets.bias.view(4, 2, 2, 2)
    assert torch.allclose(b[0, 0, 0], torch.tensor([1.0, 0.0], dtype=torch.float64))
    assert torch.allclose(b[1, :, 1], torch.tensor([0.0, 2.0], dtype=torch.float64).expand(2, 2), atol=1e-6)
    sh = torch.tensor([[4, 4], [2, 2]]); st = torch.tensor([0, 16])
    q = torch.randn(2, 5, 32, dtype=torch.float64); x = torch.randn(2, 20, 32, dtype=torch.float64)
    ref = torch.rand(2, 5, 2, 2, dtype=torch.float64)
    out = m(q, ref, x, sh, st)
    assert out.shape == (2, 5, 32)
    # zero attention weights -> uniform average of the 4 taps per head
    v = m.value_proj(x).view(2, 20, 4, 4)
    loc = ref[:, :, None, :, None, :] + m.sampling_offsets.bias.view(1, 1, 4, 2, 2, 2) / \
        torch.tensor([[4.0, 4.0], [2.0, 2.0]], dtype=torch.float64)[None, None, None, :, None, :]
    want = m.output_proj(OracleMSDAFunction.apply(v.contiguous(), sh, st, loc.expand(2, 5, 4, 2, 2, 2).contiguous(),
                                                  torch.full((2, 5, 4, 2, 2), 0.25, dtype=torch.float64), 1))
    assert torch.allclose(out, want, atol=1e-12)


ENC_CASES = ["enc_injector", "enc_extractor", "enc_boxes_padded"]


@pytest.mark.parametrize("name", ENC_CASES)
def test_encoder_ms_deform_attn_matches_reference(name, oracle_op):
    """The ViT-Adapter's MSDeformAttn (encoders/vit_adapter/ops/modules/ms_deform_attn.py:28-131) at the
    injector / extractor geometries (adapter_modules.py:108-154), plus the two branches no caller takes
    (box reference points, padding mask): output, input gradients, every parameter gradient."""
    from mmfs_amd.modules import MSDeformAttn
    z = load_golden(name)
    cfg = ast.literal_eval(str(z["cfg"]))
    m = load_params(MSDeformAttn(**cfg).double(), z)
    q = T(z["query"]).requires_grad_(True)
    f = T(z["feat"]).requires_grad_(True)
    pad = T(z["padding_mask"]) if "padding_mask" in z else None
    out = m(q, T(z["reference_points"]), f, T(z["spatial_shapes"]), T(z["level_start_index"]), pad)
    close(out, z["out"], 1e-10)
    out.backward(T(z["grad_out"]))
    close(q.grad, z["grad_query"], 1e-9)
    close(f.grad, z["grad_feat"], 1e-9)
    for k, p in m.named_parameters():
        close(p.grad, z["grad." + k], 1e-9)


# ------------------------------------------------------------------ bank builders
def test_bank_builders_match_reference():
    from mmfs_amd import bank
    z = load_golden("bank_builders")
    text_ids = T(z["text_ids"]); num = T(z["num_image_per_seq"])
    ms = [T(z[f"ms.{i}"], torch.float32) for i in range(4)]
    out = bank.prepare_mmfs_features_for_mm_decoder(text_ids, num, ms, bos_token_id=1, soi_token_id=32000,
                                                    spatial_shapes=[8, 4, 2], max_num_image=3)
    assert torch.equal(out["cross_attention_mask"], T(z["cross_attention_mask"], torch.float32))
    assert torch.equal(out["mmfs_features_mm"], T(z["mmfs_features_mm"], torch.float32))
    feats, mask = bank.prepare_mmfs_features_for_image_decoder(ms[1:], text_ids, T(z["nearest_bos_idxs"]), num,
                                                               soi_token_id=32000)
    assert torch.equal(mask, T(z["img_mask"]))
    for i, f in enumerate(feats):
        assert torch.equal(f, T(z[f"img_feat.{i}"], torch.float32))


def test_cached_tables_survive_an_inference_mode_first_call(oracle_op):
    """An evaluation pass under torch.inference_mode() before training must not poison the caches:
    the level tables, the pixel reference grid and the resized position table are kept across calls."""
    from mmfs_amd import levels
    from mmfs_amd.blocks import MMFSBlock
    from mmfs_amd.blocks import sd_mmfs
    levels._cache.clear(); sd_mmfs._ref_cache.clear()
    with contextlib.redirect_stdout(io.StringIO()):
        blk = MMFSBlock(attn_dim=32, query_dim=16, feat_dim=32, num_heads=4, n_points=2, n_levels=2,
                        gradient_checkpointing=False, grid_size=8, spatial_shapes=[4, 2],
                        base_spatial_shape=4, max_num_image_per_seq=5).double()
    with torch.no_grad():
        blk.conv.weight.normal_(0, 0.1)
    sample = torch.randn(1, 16, 4, 4, dtype=torch.float64)
    feat = torch.randn(1, 1, 20, 32, dtype=torch.float64)
    mask = torch.ones(1, 1, dtype=torch.long)
    with torch.inference_mode():
        blk(sample, feat, mask, [(4, 4), (2, 2)])
    s2, f2 = sample.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    blk(s2, f2, mask, [(4, 4), (2, 2)]).sum().backward()         # used to raise: inference tensors saved for backward
    assert s2.grad is not None and f2.grad is not None


def test_level_tables_are_cached_and_marked_canonical():
    import MultiScaleDeformableAttention as MSDA
    from mmfs_amd.levels import make_level_tables
    a = make_level_tables([(8, 8), (4, 4)], 3, "cpu")
    b = make_level_tables([(8, 8), (4, 4)], 3, "cpu")
    assert a[0] is b[0] and a[2] == 240
    assert a[0].tolist() == [[8, 8], [4, 4]] * 3 and a[1].tolist() == [0, 64, 80, 144, 160, 224]
    assert MSDA.levels_are_canonical(a[0], a[1], 240)


def test_bank_from_levels_equals_pack_then_gather():
    """llm_feature_bank_from_levels (the one-pass route; framework ops on host tensors) builds the
    same bank as packing the levels and gathering per sequence, padded slots zero."""
    from mmfs_amd import bank
    g = torch.Generator().manual_seed(5)
    levels = [torch.randn(6, 8, s, s, generator=g, dtype=torch.float64) for s in (4, 2, 1)]
    num = torch.tensor([2, 0, 3, 1])
    a = bank.llm_feature_bank_from_levels(levels, num, 3)
    b = bank.llm_feature_bank(bank.pack_image_levels(levels), num, 3)
    assert a.shape == (4, 3, 21, 8) and torch.equal(a, b)
    assert float(a[1].abs().max()) == 0.0 and float(a[0, 2].abs().max()) == 0.0
    src = torch.tensor([5, -1, 0, 9])
    got = bank.gather_bank(levels, src)
    assert torch.equal(got[0], bank.pack_image_levels(levels)[5]) and float(got[1].abs().max()) == 0.0
    assert float(got[3].abs().max()) == 0.0          # past the last image: an empty slot, not a clamp


def _llama_stack(n_layers, seed=0):
    from mmfs_amd.blocks import LlamaMMFSAttention
    cfg = types.SimpleNamespace(hidden_size=64, num_attention_heads=4, rms_norm_eps=1e-6,
                                max_position_embeddings=64, image_embed_dim=32, spatial_shapes=[8, 4, 2])
    g = torch.Generator().manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        layers = [LlamaMMFSAttention(cfg, layer_idx=4 * i).double() for i in range(n_layers)]
    with torch.no_grad():
        for l in layers:                                    # nothing at its (zero / one) initial value
            for p in l.parameters():
                if p.requires_grad:
                    p.copy_(torch.randn(p.shape, generator=g, dtype=torch.float64) * 0.3 + (1.0 if p.dim() == 1 and p.numel() > 1 else 0.0))
    return layers


def test_llama_schedule_equals_the_references_layer_by_layer(oracle_op):
    """The decoder's MMFS layers all normalise and project the SAME bank (modeling_llama_mmfs.py:352-353, 581-583):
    one RMS pass with the gains folded into the projections and one batched GEMM (LlamaMMFSSchedule) give the
    outputs and EVERY gradient of the reference's schedule -- norm2 + value_proj inside each layer."""
    from mmfs_amd.blocks import LlamaMMFSSchedule
    layers = _llama_stack(3)
    g = torch.Generator().manual_seed(1)
    B, Lq, n, hw = 2, 5, 2, 64 + 16 + 4
    hidden = torch.randn(B, Lq, 64, generator=g, dtype=torch.float64)
    feats = torch.randn(B, n, hw, 32, generator=g, dtype=torch.float64)
    mask = torch.tensor([[[1, 1]] * Lq, [[1, 0]] * Lq], dtype=torch.float32)
    go = torch.randn(B, Lq, 64, generator=g, dtype=torch.float64)

    def run(fused):
        for l in layers:
            l.zero_grad()
        h0 = hidden.clone().requires_grad_(True); f0 = feats.clone().requires_grad_(True)
        bank = LlamaMMFSSchedule(layers).project(f0) if fused else None
        h = h0
        for k, l in enumerate(layers):
            h = h + (l(h, f0, mask, value=bank.values[k]) if fused else l(h, f0, mask))
        h.backward(go)
        return h.detach(), h0.grad, f0.grad, {n_: p.grad.clone() for l in layers for n_, p in l.named_parameters() if p.grad is not None}

    ref, fus = run(False), run(True)
    close(fus[0], ref[0].numpy(), 1e-10)
    close(fus[1], ref[1].numpy(), 1e-10)
    # (the gradient w.r.t. the bank passes through the norm's fp32 statistics -- the reference computes them in
    # fp32 whatever the input type, modeling_llama_mmfs.py:61-63 -- once here, once per layer there: fp32 rounding)
    close(fus[2], ref[2].numpy(), 1e-6)
    assert ref[3].keys() == fus[3].keys() and len(ref[3]) >= 12
    for k in ref[3]:
        close(fus[3][k], ref[3][k].numpy(), 1e-9)


def test_llama_schedule_keeps_the_projected_bank_only_while_nothing_moves():
    """Outside autograd (generation: the bank is constant across decode steps, mm_interleaved.py:598-664) the
    projections are kept for the same, unmodified bank tensor and parameters; anything else recomputes."""
    from mmfs_amd.blocks import LlamaMMFSSchedule
    layers = _llama_stack(2, seed=3)
    for l in layers:
        l.eval()                                                           # (nothing is kept in training mode: ADVICE r3)
    sched = LlamaMMFSSchedule(layers)
    feats = torch.randn(1, 1, 84, 32, dtype=torch.float64)
    with torch.no_grad():
        a = sched.project(feats)
        assert sched.project(feats) is a                                   # same tensor, same version: kept
        feats.mul_(2.0)
        b = sched.project(feats)
        assert b is not a                                                  # modified in place: recomputed
        layers[1].norm2.weight.add_(0.1)
        c = sched.project(feats)
        assert c is not b and not torch.equal(c.values[1], b.values[1]) and torch.equal(c.values[0], b.values[0])
        assert sched.project(feats.clone()) is not c                       # another tensor object
    d = sched.project(feats)                                               # with autograd: never kept
    assert d is not c and sched._projected is None
    for k, l in enumerate(layers):
        assert torch.allclose(d.values[k], l.attn.value_proj(l.norm2(feats)), atol=1e-12)


def test_decode_caches_follow_the_parameters_and_the_mask(oracle_op):
    """Without gradients (sampling, decoding) a layer keeps what depends on its PARAMETERS only -- the relative-position
    table pushed through the two heads, tanh(gate) -- and the layers of a step can share the images' ranks
    (``LlamaMMFSSchedule.image_ranks``: a function of the mask only).  Same outputs as with gradients enabled (where
    nothing is kept); a parameter that moves in place is seen."""
    from mmfs_amd.blocks import LlamaMMFSSchedule
    layers = _llama_stack(2, seed=5)
    for l in layers:
        l.eval()
    g = torch.Generator().manual_seed(2)
    B, Lq, n, hw = 2, 3, 2, 64 + 16 + 4
    hidden = torch.randn(B, Lq, 64, generator=g, dtype=torch.float64)
    feats = torch.randn(B, n, hw, 32, generator=g, dtype=torch.float64)
    mask = torch.tensor([[[1, 1]] * Lq, [[1, 0]] * Lq], dtype=torch.float32)

    sched = LlamaMMFSSchedule(layers)

    def stack(m, shared=False):
        ranks = sched.image_ranks(m, Lq) if shared else None
        h = hidden
        for l in layers:
            h = h + l(h, feats, m, image_ranks=ranks)
        return h

    want = stack(mask.clone()).detach()                                  # gradients enabled: nothing kept
    assert all(l.attn._tables is None for l in layers)
    with torch.no_grad():
        a = stack(mask)
        t0 = layers[0].attn._tables
        assert t0 is not None
        b = stack(mask, shared=True)
        assert layers[0].attn._tables is t0                              # second step: kept
        close(a, want.numpy(), 1e-12); close(b, want.numpy(), 1e-12)
        # a parameter moves in place (an optimiser step between two evaluations): seen
        layers[0].attn.query_relpos.weight.add_(0.5)
        layers[1].gate.add_(0.25)
        c = stack(mask)
        assert layers[0].attn._tables is not t0
    want2 = stack(mask.clone()).detach()
    close(c, want2.numpy(), 1e-12)
    assert not torch.allclose(c, a)
    with torch.no_grad():
        mask[1, :, 1] = 1.0                                              # (the second sequence's second image becomes visible)
        d = stack(mask, shared=True)
    close(d, stack(mask.clone()).detach().numpy(), 1e-12)
    assert not torch.allclose(d, c)


def test_folded_query_projection_is_the_two_gemms():
    """Without gradients ``MMFS._plan_tables`` keeps the stacked heads' weights already multiplied with
    ``dynamic_offset_mask`` (nothing non-linear stands between them): one GEMM on the query itself.  The kept product
    equals the two-GEMM statement (fp64: to rounding), follows every parameter that went into it, is not made with
    gradients enabled, and ``fold_query_projection = False`` keeps the two GEMMs."""
    from mmfs_amd.modules import MMFS
    torch.manual_seed(4)
    with contextlib.redirect_stdout(io.StringIO()):
        m = MMFS(d_model=32, d_query=24, d_value=16, d_out=24, n_levels=2, n_heads=4, n_points=4, ratio=1.0,
                 spatial_shapes=[8, 4], base_spatial_shape=4, max_num_image_per_seq=6).double().eval()
    with torch.no_grad():
        m.sampling_offsets.weight.normal_(0, 0.1)
        m.attention_weights.weight.normal_(0, 0.1)
        m.dynamic_offset_mask.bias.normal_(0, 0.1)
    x = torch.randn(3, 5, 24, dtype=torch.float64)
    H, L, P = 4, 2, 4
    with torch.no_grad():
        off_tab, att_tab, aw_w, aw_b, cat_w, cat_b, fold_w, fold_b = m._plan_tables(True)
        assert fold_w.shape == (H * P * 2 + H * L * P, 24) and cat_w.shape == fold_w.shape
        want = torch.nn.functional.linear(m.dynamic_offset_mask(x), cat_w, cat_b)
        got = torch.nn.functional.linear(x, fold_w, fold_b)
        assert float((got - want).abs().max()) <= 1e-12
        # the two column ranges are the two heads (the attention head without its sink columns)
        assert torch.allclose(want[..., :H * P * 2], m.sampling_offsets(m.dynamic_offset_mask(x)), atol=1e-13)
        full = m.attention_weights(m.dynamic_offset_mask(x)).view(3, 5, H, L, P + 1)[..., :P].reshape(3, 5, -1)
        assert torch.allclose(want[..., H * P * 2:], full, atol=1e-13)
        assert m._plan_tables(True)[6] is fold_w                          # kept
        m.dynamic_offset_mask.weight.mul_(1.5)                             # a parameter of the product moves in place
        again = m._plan_tables(True)
        assert again[6] is not fold_w
        assert float((torch.nn.functional.linear(x, again[6], again[7])
                      - torch.nn.functional.linear(m.dynamic_offset_mask(x), again[4], again[5])).abs().max()) <= 1e-12
        m.fold_query_projection = False
        assert m._plan_tables(True)[6] is None and m._plan_tables(True)[4] is not None
        m.fold_query_projection = True
        # a process-wide module hook (a profiler's, a FLOP counter's) must see dynamic_offset_mask run: no fold while
        # it is registered -- and the tables made then must not be the ones used after it is gone
        assert m._plan_tables(True)[6] is not None
        handle = torch.nn.modules.module.register_module_forward_hook(lambda mod, i, o: None)
        try:
            assert m._plan_tables(True)[6] is None
        finally:
            handle.remove()
        assert m._plan_tables(True)[6] is not None
    # with gradients: nothing kept, nothing folded; the heads' weights stacked per call (part of the graph), the two
    # tables one GEMM on them
    t = m._plan_tables(True)
    assert t[6] is None and t[4].requires_grad and t[1] is None and t[0].shape == (6, H * P * 2 + H * L * P)
    assert m._tables is None or m._tables[1] is not t
    m.stack_heads_in_training = False
    t = m._plan_tables(True)
    assert t[4] is None and t[6] is None and t[1] is not None
    m.stack_heads_in_training = True


def test_folded_linear_is_the_two_layers_and_follows_its_parameters():
    """``FoldedLinear`` (what ``MMFSBlock`` does with output_proj + its 1x1 convolution and ``LlamaMMFSAttention`` with
    output_proj + tanh(gate) when no gradient is wanted): the kept product equals the two layers to rounding (fp64), is
    kept while nothing moves, follows a parameter that moves in place; and the blocks' no-grad outputs are their
    with-grad ones."""
    from mmfs_amd.modules.mmfs import FoldedLinear
    g = torch.Generator().manual_seed(1)
    wi, bi = torch.randn(7, 5, generator=g, dtype=torch.float64), torch.randn(7, generator=g, dtype=torch.float64)
    wo, bo = torch.randn(3, 7, generator=g, dtype=torch.float64), torch.randn(3, generator=g, dtype=torch.float64)
    x = torch.randn(4, 5, generator=g, dtype=torch.float64)
    F_ = torch.nn.functional
    fl = FoldedLinear()
    w, b = fl.get(wi, bi, wo, bo)
    assert float((F_.linear(x, w, b) - F_.linear(F_.linear(x, wi, bi), wo, bo)).abs().max()) <= 1e-13
    assert fl.get(wi, bi, wo, bo)[0] is w
    wo.mul_(2.0)
    w2, b2 = fl.get(wi, bi, wo, bo)
    assert w2 is not w and float((F_.linear(x, w2, b2) - F_.linear(F_.linear(x, wi, bi), wo, bo)).abs().max()) <= 1e-13
    gate = torch.tensor([0.3], dtype=torch.float64).tanh()
    wg, bg = FoldedLinear().get(wi, bi, gate, None)
    assert float((F_.linear(x, wg, bg) - F_.linear(x, wi, bi) * gate).abs().max()) <= 1e-14
    wn, bn = FoldedLinear().get(wi, None, wo, bo)                      # an inner layer without bias
    assert float((F_.linear(x, wn, bn) - F_.linear(F_.linear(x, wi), wo, bo)).abs().max()) <= 1e-13


def test_blocks_without_gradients_equal_blocks_with(oracle_op):
    """The folds the blocks take when no gradient is wanted (conv into output_proj, gate into output_proj) against the
    same call with gradients enabled (nothing folded): fp64, to rounding."""
    z = load_golden("block_sd_mmfs_net")
    net = _tiny_net(z).eval()
    res = [T(z[f"res.{i}"]) for i in range(6)]
    feats = [T(z[f"feat.{i}"]) for i in range(3)]
    a = net(T(z["mid"]), res, feats, T(z["ms_mask"]))
    with torch.no_grad():
        b = net(T(z["mid"]), res, feats, T(z["ms_mask"]))
        assert net.mmfs_mid_block._conv_fold._kept is not None          # the fold was taken
    close(b[0], a[0].detach().numpy(), 1e-11)
    for x, y in zip(b[1], a[1]):
        close(x, y.detach().numpy(), 1e-11)
    layers = [l.eval() for l in _llama_stack(2, seed=3)]
    h = torch.randn(2, 5, layers[0].hidden_size, dtype=torch.float64)
    f = torch.randn(2, 1, 84, 32, dtype=torch.float64)
    mask = torch.ones(2, 5, 1, dtype=torch.float64)
    with torch.no_grad():
        for l in layers:
            l.gate.fill_(0.7)
    want = layers[1](h, f, mask)
    with torch.no_grad():
        got = layers[1](h, f, mask)
        assert layers[1]._gate_fold._kept is not None
    close(got, want.detach().numpy(), 1e-12)


def test_llama_layer_in_inference_mode_then_no_grad_then_training(oracle_op):
    """The folds and kept tensors of a layer's no-grad path under ``torch.inference_mode()`` (where the kept tanh(gate)
    is an inference tensor: no version counter), then under ``no_grad``, then with gradients: the same output each time,
    and a gate that moves in place afterwards is seen."""
    layers = _llama_stack(1, seed=5)
    l = layers[0]
    h = torch.randn(2, 3, l.hidden_size, dtype=torch.float64)
    f = torch.randn(2, 1, 84, 32, dtype=torch.float64)
    mask = torch.ones(2, 3, 1, dtype=torch.float64)
    with torch.no_grad():
        l.gate.fill_(0.4)
    with torch.inference_mode():
        a = l(h, f, mask).clone()
        a2 = l(h, f, mask)                                  # second call: everything kept
        assert torch.equal(a, a2)
    with torch.no_grad():
        b = l(h, f, mask)
    c = l(h, f, mask)
    close(b, a.numpy(), 1e-12)
    close(c.detach(), a.numpy(), 1e-12)
    c.sum().backward()
    assert l.gate.grad is not None and float(l.gate.grad.abs().sum()) > 0
    with torch.no_grad():
        l.gate.fill_(0.8)
        d = l(h, f, mask)
    assert float((d - b).abs().max()) > 1e-6
    close(d, (l(h, f, mask)).detach().numpy(), 1e-12)
    # ``residual=``: the decoder layer's own next statement, with and without gradients
    with torch.no_grad():
        close(l(h, f, mask, residual=h), (h + d).numpy(), 1e-12)
    close(l(h, f, mask, residual=h).detach(), (h + d).numpy(), 1e-12)


@pytest.mark.parametrize("with_bias", [True, False])
def test_gated_projection_function_gradients(with_bias):
    """``GatedProjectionFunction`` (residual + g (x W^T + b), the gate applied to the small side of every backward product)
    against autograd's gradients of the framework statement, fp64."""
    from mmfs_amd.functions.block_func import GatedProjectionFunction
    g = torch.Generator().manual_seed(2)
    mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).requires_grad_(True)
    x, W, b, gate, res = mk(2, 5, 6), mk(7, 6), (mk(7) if with_bias else None), mk(1), mk(2, 5, 7)
    go = torch.randn(2, 5, 7, generator=g, dtype=torch.float64)
    leaves = [t for t in (x, W, b, gate, res) if t is not None]
    got = torch.autograd.grad(GatedProjectionFunction.apply(x, W, b, gate.tanh(), res), leaves, go)
    want = torch.autograd.grad(res + torch.nn.functional.linear(x, W, b) * gate.tanh(), leaves, go)
    for a, w in zip(got, want):
        close(a, w.numpy(), 1e-12)


@pytest.mark.parametrize("token_grad", [False, True])
def test_ignore_token_function_gradients(token_grad):
    """``IgnoreTokenFunction`` (out + token * sink per head as one block-diagonal product each way) against autograd's
    gradients of the framework statement, fp64."""
    from mmfs_amd.functions.block_func import IgnoreTokenFunction
    g = torch.Generator().manual_seed(3)
    T, H, D = 11, 4, 6
    out = torch.randn(T, H * D, generator=g, dtype=torch.float64).requires_grad_(True)
    tok = torch.randn(H, D, generator=g, dtype=torch.float64).requires_grad_(token_grad)
    sink = torch.rand(T, H, generator=g, dtype=torch.float64).requires_grad_(True)
    go = torch.randn(T, H * D, generator=g, dtype=torch.float64)
    leaves = [out, sink] + ([tok] if token_grad else [])
    y = IgnoreTokenFunction.apply(out, tok, sink)
    want_y = out + (tok[None] * sink[..., None]).reshape(T, H * D)
    close(y.detach(), want_y.detach().numpy(), 1e-12)
    for a, w in zip(torch.autograd.grad(y, leaves, go), torch.autograd.grad(want_y, leaves, go)):
        close(a, w.numpy(), 1e-12)


def test_mmfs_net_in_inference_mode_then_training(oracle_op):
    """The same for the image decoder's net: a first call inside ``torch.inference_mode()`` (kept projections, folded
    convolutions, position tables made there) must not poison a later ``no_grad`` call or a training step."""
    z = load_golden("block_sd_mmfs_net")
    net = _tiny_net(z)
    res = [T(z[f"res.{i}"]) for i in range(6)]
    feats = [T(z[f"feat.{i}"]) for i in range(3)]
    with torch.inference_mode():
        a = net(T(z["mid"]), res, feats, T(z["ms_mask"]))
        a = (a[0].clone(), [r.clone() for r in a[1]])
    with torch.no_grad():
        b = net(T(z["mid"]), res, feats, T(z["ms_mask"]))
    mid = T(z["mid"]).requires_grad_(True)
    c = net(mid, res, feats, T(z["ms_mask"]))
    c[0].sum().backward()
    assert mid.grad is not None
    for x, y, w in zip([a[0]] + a[1], [b[0]] + list(b[1]), [c[0]] + list(c[1])):
        close(y, x.numpy(), 1e-11)
        close(w.detach(), x.numpy(), 1e-11)


def test_feature_tensors_made_in_inference_mode_are_accepted(oracle_op):
    """A pipeline that runs wholly inside ``torch.inference_mode()`` hands over feature tensors that have no version
    counter (reading it raises): the identity caches of the kept projections take them (same object = same bank), for
    the image decoder's net and for the LLM-side schedule."""
    from mmfs_amd.blocks import LlamaMMFSSchedule
    z = load_golden("block_sd_mmfs_net")
    net = _tiny_net(z).eval()
    layers = [l.eval() for l in _llama_stack(2, seed=3)]
    with torch.inference_mode():
        res = [T(z[f"res.{i}"]) * 1.0 for i in range(6)]               # inference tensors
        feats = [T(z[f"feat.{i}"]) * 1.0 for i in range(3)]
        assert feats[0].is_inference()
        a = net(T(z["mid"]) * 1.0, res, feats, T(z["ms_mask"]))
        kept = net.__dict__.get("_projected")
        b = net(T(z["mid"]) * 1.0, res, feats, T(z["ms_mask"]))
        assert kept is not None and net.__dict__.get("_projected") is kept      # the same feature objects: kept
        assert torch.equal(a[0], b[0])
        net(T(z["mid"]) * 1.0, res, [f.clone() for f in feats], T(z["ms_mask"]))
        assert net.__dict__.get("_projected") is not kept                        # other objects: recomputed
        sched = LlamaMMFSSchedule(layers)
        bank = torch.randn(1, 1, 84, 32, dtype=torch.float64) * 1.0
        p1 = sched.project(bank)
        assert sched.project(bank) is p1 and sched.project(bank.clone()) is not p1



# ---------------------------------------------------------------------------------------------------------
# the kept no-grad artefacts were keyed on (data pointer, version counter) of their parameters
# only, and a write through ``param.data`` moves neither -- DeepSpeed's bit16 update, EMAModel.copy_to.
def test_kept_folds_do_not_survive_a_mode_change_a_state_dict_load_or_an_invalidation(oracle_op):
    """The advisor's reproduction, and the three ways out: (1) nothing is kept in training mode, and train() / eval()
    drop whatever was kept before (DeepSpeed's periodic evaluation switches modes around every update); (2) loading a
    state dict drops it; (3) ``clear_caches()`` / ``mmfs_amd.invalidate_caches()`` for writes through ``.data`` while
    the modules stay in eval mode.  After each, the no-grad output is the with-grad (nothing kept) output."""
    import mmfs_amd
    from mmfs_amd.blocks import LlamaMMFSSchedule
    layers = [l.eval() for l in _llama_stack(2, seed=7)]
    l = layers[0]
    g = torch.Generator().manual_seed(0)
    h = torch.randn(2, 4, l.hidden_size, generator=g, dtype=torch.float64)
    f = torch.randn(2, 2, 84, 32, generator=g, dtype=torch.float64)
    mask = torch.ones(2, 4, 2, dtype=torch.float64)

    def nograd():
        with torch.no_grad():
            return l(h, f, mask)

    def truth():
        return l(h, f, mask).detach()                                     # with gradients nothing is kept or folded

    a = nograd()
    close(a, truth().numpy(), 1e-12)
    assert l._gate_fold._kept is not None and l.attn._tables is not None          # (the folds were taken)
    # --- the hazard itself: a write through .data is invisible to the counters ...
    l.gate.data.fill_(1.0)
    l.attn.query_relpos.weight.data.mul_(-2.0)
    l.attn.output_proj.weight.data.mul_(0.5)
    stale = nograd()
    assert float((stale - truth()).abs().max()) > 1e-3                    # ... and eval-mode folds DO go stale (documented)
    # (3) the public invalidation
    l.clear_caches()
    close(nograd(), truth().numpy(), 1e-12)
    l.gate.data.fill_(0.3)
    mmfs_amd.invalidate_caches()
    close(nograd(), truth().numpy(), 1e-12)
    # (1) a training step between two evaluations, parameters updated through .data as DeepSpeed does
    l.train()
    l.gate.data.fill_(-0.8)
    l.attn.dynamic_offset_mask.weight.data.mul_(1.5)
    in_train = nograd()                                                   # no_grad inside training mode: nothing kept
    close(in_train, truth().numpy(), 1e-12)
    l.eval()
    close(nograd(), truth().numpy(), 1e-12)
    l.train(); l.attn.sampling_offsets.bias.data.add_(0.25); l.eval()
    close(nograd(), truth().numpy(), 1e-12)
    # (2) a state dict loaded in eval mode (load_state_dict copies under no_grad; here through .data to be sure)
    sd = {k: v.clone() * 0.5 for k, v in l.state_dict().items()}
    nograd()
    l.load_state_dict(sd)
    close(nograd(), truth().numpy(), 1e-12)
    # the schedule's projected bank and the image decoder's net follow the same rule
    sched = LlamaMMFSSchedule(layers)
    with torch.no_grad():
        p1 = sched.project(f)
        assert sched.project(f) is p1
        layers[1].attn.value_proj.weight.data.mul_(2.0)
        assert sched.project(f) is p1                                     # (the hazard)
        sched.clear_cache()
        p2 = sched.project(f)
        assert p2 is not p1 and not torch.equal(p2.values[1], p1.values[1])
        layers[1].train(); layers[1].eval()
        assert sched.project(f) is not p2
    z = load_golden("block_sd_mmfs_net")
    net = _tiny_net(z).eval()
    res = [T(z[f"res.{i}"]) for i in range(6)]
    feats = [T(z[f"feat.{i}"]) for i in range(3)]
    with torch.no_grad():
        net(T(z["mid"]), res, feats, T(z["ms_mask"]))
        assert net.__dict__.get("_projected") is not None
        net.mmfs_mid_block.conv.weight.data.mul_(3.0)
        net.mmfs_mid_block.mmfs.value_proj.weight.data.mul_(0.5)
        net.clear_feature_cache()
        b = net(T(z["mid"]), res, feats, T(z["ms_mask"]))
    want = net(T(z["mid"]), res, feats, T(z["ms_mask"]))
    close(b[0], want[0].detach().numpy(), 1e-11)
    net.train()
    with torch.no_grad():
        net(T(z["mid"]), res, feats, T(z["ms_mask"]))
    assert net.__dict__.get("_projected") is None                         # nothing kept in training mode
    kept = net.mmfs_mid_block._conv_fold._kept
    assert kept is None or kept[0][0] != mmfs_amd.levels.cache_epoch()    # ... and the fold of the eval pass is out of date


def test_hooked_layers_are_called_as_layers(oracle_op):
    """the no-grad paths evaluate ``output_proj`` / ``dynamic_offset_mask`` / the 1x1 convolution
    without calling the modules only when NO hook could observe the call -- forward or backward, the layer's own or a
    process-wide one -- and never with gradients enabled."""
    from mmfs_amd.levels import hook_free
    layers = [l.eval() for l in _llama_stack(1, seed=2)]
    l = layers[0]
    h = torch.randn(1, 3, l.hidden_size, dtype=torch.float64)
    f = torch.randn(1, 1, 84, 32, dtype=torch.float64)
    mask = torch.ones(1, 3, 1, dtype=torch.float64)
    proj = l.attn.output_proj
    assert hook_free(proj)
    seen = []
    hb = proj.register_full_backward_hook(lambda m, gi, go: seen.append("bwd"))
    assert not hook_free(proj)
    out = l(h.requires_grad_(True), f, mask)
    out.sum().backward()
    assert seen == ["bwd"]                                                # with gradients the layer is called as a layer
    hb.remove()
    hg = torch.nn.modules.module.register_module_forward_hook(lambda m, i, o: seen.append(type(m).__name__) if m is proj else None)
    try:
        assert not hook_free(proj)
        with torch.no_grad():
            l(h.detach(), f, mask)
        assert "Linear" in seen                                           # a global hook sees output_proj's call
    finally:
        hg.remove()
    assert hook_free(proj)
    # the decoder-level schedule (one batched value projection for all layers) steps aside for a hooked value_proj
    from mmfs_amd.blocks import LlamaMMFSSchedule
    sched = LlamaMMFSSchedule(layers)
    assert sched.can_fuse()
    hv = l.attn.value_proj.register_forward_hook(lambda m, i, o: seen.append("value_proj"))
    try:
        assert not sched.can_fuse()
        with torch.no_grad():
            bank = sched.project(f)
            l(h.detach(), f, mask, value=bank.values[0])
        assert "value_proj" in seen
    finally:
        hv.remove()
    assert sched.can_fuse()
