#!/usr/bin/env python3
"""tests/golden/make_golden.py -- generates the committed golden vectors.

Runs ONLY in the build container, where the reference is mounted read-only at
/root/reference.  It imports the reference's Python code (nothing is copied),
feeds it seeded inputs and stores inputs + outputs as small ``.npz`` fixtures
next to this script.  The GPU box has no /root/reference: tests there read the
fixtures only.

What is exercised in the reference (the only CPU-runnable statements of the path):
  * ``ms_deform_attn_core_pytorch``    ops/functions/ms_deform_attn_func.py:47-67
      forward + autograd backward, fp64 and fp32                     -> op_*.npz
  * ``MMFS.forward``                   ops/modules/mmfs.py:120-276
      with its ``MSDeformAttnFunction`` name bound to the function above (the
      native extension cannot be built here)                          -> mmfs_*.npz
  * ``LlamaMMFSAttention``             decoders/modeling_llama_mmfs.py:311-367
    ``MMFSBlock`` / ``MMFSNet``        decoders/sd_mmfs.py:44-272    -> block_*.npz
  * ``MSDeformAttn`` (encoder twin)    encoders/vit_adapter/ops/modules/ms_deform_attn.py:28-131 -> enc_*.npz
  * feature-bank builders              mm_interleaved.py:185-252, 306-340
      (the two methods are compiled from the file's AST without importing the
      module, whose imports need diffusers/timm/...)                  -> bank_*.npz

Input distributions follow the reference's own test scripts
(ops/tests/forward_backward_error.py:28-47, ops/tests/create_data.py:11-30):
value, loc ~ U[0,1); attn ~ U[0,1)+1e-5 normalised over (L,P); seed 0.

Usage:  python tests/golden/make_golden.py     (rewrites the .npz files)
"""
import ast
import importlib
import os
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
UTILS = os.path.join(REF, "mm_interleaved/models/utils")


# ----------------------------------------------------------------------------- reference imports
def import_reference_ops():
    sys.path.insert(0, UTILS)
    funcs = importlib.import_module("ops.functions.ms_deform_attn_func")
    mods = importlib.import_module("ops.modules.mmfs")

    class CoreAsFunction:
        """6-argument adapter so MMFS can call the reference's pure-PyTorch core."""
        @staticmethod
        def apply(value, shapes, start, loc, attn, im2col_step):
            return funcs.ms_deform_attn_core_pytorch(value, shapes, loc, attn)

    mods.MSDeformAttnFunction = CoreAsFunction
    return funcs, mods


def import_reference_blocks():
    """Package shells so that decoders/*.py import without models/__init__.py."""
    for name, rel in [("mm_interleaved", "mm_interleaved"),
                      ("mm_interleaved.models", "mm_interleaved/models"),
                      ("mm_interleaved.models.decoders", "mm_interleaved/models/decoders"),
                      ("mm_interleaved.models.utils", "mm_interleaved/models/utils")]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF, rel)]
            sys.modules[name] = m
    core = importlib.import_module("mm_interleaved.models.utils.ops.functions.ms_deform_attn_func")
    mmfs = importlib.import_module("mm_interleaved.models.utils.ops.modules.mmfs")

    class CoreAsFunction:
        @staticmethod
        def apply(value, shapes, start, loc, attn, im2col_step):
            return core.ms_deform_attn_core_pytorch(value, shapes, loc, attn)

    mmfs.MSDeformAttnFunction = CoreAsFunction
    sd = importlib.import_module("mm_interleaved.models.decoders.sd_mmfs")
    llama = importlib.import_module("mm_interleaved.models.decoders.modeling_llama_mmfs")
    return sd, llama


def load_reference_methods(path, names):
    """Compile selected method definitions out of a reference file's AST."""
    tree = ast.parse(open(path).read())
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            ns = {"torch": torch, "rearrange": importlib.import_module("einops").rearrange,
                  "Optional": __import__("typing").Optional, "List": __import__("typing").List}
            exec(compile(mod, path, "exec"), ns)
            found[node.name] = ns[node.name]
    return found


# ----------------------------------------------------------------------------- helpers
def level_tables(shapes):
    sh = torch.as_tensor(shapes, dtype=torch.long)
    start = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    return sh, start


def op_inputs(B, H, D, Nq, P, shapes, loc_lo=0.0, loc_hi=1.0, gen=None):
    """The reference test distribution (create_data.py:11-30), drawn in fp32."""
    sh, start = level_tables(shapes)
    S = int(sh.prod(1).sum())
    L = len(shapes)
    value = torch.rand(B, S, H, D, generator=gen)
    loc = torch.rand(B, Nq, H, L, P, 2, generator=gen) * (loc_hi - loc_lo) + loc_lo
    attn = torch.rand(B, Nq, H, L, P, generator=gen) + 1e-5
    attn = attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    return value, sh, start, loc, attn


def run_core(core, value, sh, loc, attn, grad_out, dtype):
    v = value.to(dtype).requires_grad_(True)
    l = loc.to(dtype).requires_grad_(True)
    a = attn.to(dtype).requires_grad_(True)
    out = core(v, sh, l, a)
    out.backward(grad_out.to(dtype).reshape(out.shape))
    return out.detach(), v.grad, l.grad, a.grad


def save(name, **arrays):
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        v = np.asarray(v)
        if v.dtype == np.float64 and np.array_equal(v.astype(np.float32).astype(np.float64), v):
            v = v.astype(np.float32)          # lossless: inputs/params are drawn in fp32
        conv[k] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def op_case(core, name, *, B, H, D, Nq, P, shapes, seed, loc_lo=0.0, loc_hi=1.0,
            grad="ones", big=False):
    gen = torch.Generator().manual_seed(seed)
    value, sh, start, loc, attn = op_inputs(B, H, D, Nq, P, shapes, loc_lo, loc_hi, gen)
    if grad == "ones":
        g = torch.ones(B, Nq, H * D)
    else:
        g = torch.randn(B, Nq, H * D, generator=gen)
    o64, gv64, gl64, ga64 = run_core(core, value, sh, loc, attn, g, torch.float64)
    o32, gv32, gl32, ga32 = run_core(core, value, sh, loc, attn, g, torch.float32)
    f = (lambda t: t.float()) if big else (lambda t: t)      # keep big fixtures small
    save(name, value=value, spatial_shapes=sh, level_start_index=start, loc=loc, attn=attn,
         grad_out=g, out_f64=o64, grad_value_f64=f(gv64), grad_loc_f64=gl64, grad_attn_f64=ga64,
         out_f32=o32, grad_value_f32=gv32 if not big else np.zeros(0, np.float32),
         grad_loc_f32=gl32, grad_attn_f32=ga32)


# ----------------------------------------------------------------------------- G1-G3: the op
def make_op_goldens(core):
    ref_shapes = [(6, 4), (3, 2)]
    # G1: forward_backward_error.py:28-47  (N=1, M=2, D in {4,64}, Lq=2, L=2, P=2)
    op_case(core, "op_g1_d4", B=1, H=2, D=4, Nq=2, P=2, shapes=ref_shapes, seed=0)
    op_case(core, "op_g1_d64", B=1, H=2, D=64, Nq=2, P=2, shapes=ref_shapes, seed=0)
    # G2: create_data.py:11-30 with bs=4, random upstream gradient
    op_case(core, "op_g2_bs4", B=4, H=2, D=4, Nq=2, P=2, shapes=ref_shapes, seed=1, grad="randn")
    # G3: reduced real geometries, locations partly outside [0,1]
    one = [(8, 8), (4, 4), (2, 2)]
    op_case(core, "op_g3_llm_n1", B=1, H=16, D=64, Nq=16, P=8, shapes=one, seed=2,
            loc_lo=-0.25, loc_hi=1.25, grad="randn", big=True)
    rect = [(6, 8), (3, 4), (2, 2)]
    op_case(core, "op_g3_rect_n3", B=2, H=4, D=32, Nq=16, P=8, shapes=rect * 3, seed=3,
            loc_lo=-0.25, loc_hi=1.25, grad="randn", big=True)
    sd = [(8, 8), (4, 4), (2, 2), (1, 1)]
    op_case(core, "op_g3_sd_n4", B=2, H=4, D=32, Nq=16, P=4, shapes=sd * 4, seed=4,
            loc_lo=-0.1, loc_hi=1.1, grad="randn", big=True)
    # odd head widths (scalar code path of any implementation)
    op_case(core, "op_g3_d24", B=2, H=3, D=24, Nq=5, P=3, shapes=[(5, 7), (2, 3)], seed=5,
            loc_lo=-0.2, loc_hi=1.2, grad="randn")
    # hand-placed pixel-centre / border locations (away from the measure-zero -1/W edges)
    sh, start = level_tables([(4, 4), (2, 3)])
    pts = torch.tensor([[0.125, 0.125], [0.5, 0.5], [0.0, 0.0], [1.0, 1.0], [0.999, 0.001],
                        [-0.05, 0.3], [0.3, 1.05], [0.875, 0.625]])
    B, H, D, Nq, L, P = 1, 2, 8, 4, 2, 2
    gen = torch.Generator().manual_seed(6)
    value = torch.rand(B, int(sh.prod(1).sum()), H, D, generator=gen)
    idx = torch.arange(B * Nq * H * L * P) % pts.shape[0]
    loc = pts[idx].reshape(B, Nq, H, L, P, 2).clone()
    attn = torch.rand(B, Nq, H, L, P, generator=gen)
    g = torch.randn(B, Nq, H * D, generator=gen)
    o64, gv64, gl64, ga64 = run_core(core, value, sh, loc, attn, g, torch.float64)
    save("op_g3_border", value=value, spatial_shapes=sh, level_start_index=start, loc=loc, attn=attn,
         grad_out=g, out_f64=o64, grad_value_f64=gv64, grad_loc_f64=gl64, grad_attn_f64=ga64)


# ----------------------------------------------------------------------------- G4: MMFS.forward
def randomise(module, gen, scale=0.2):
    """The reference zero-initialises several weights; a golden of that pins nothing."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("ignore_token"):
                new = torch.randn(p.shape, generator=gen) * scale       # frozen but non-zero
            elif p.dim() >= 2:
                new = torch.randn(p.shape, generator=gen) * (scale / max(1.0, p.shape[-1] ** 0.5) * 4)
            else:
                new = torch.randn(p.shape, generator=gen) * scale
            p.copy_(new.float())            # fp32-representable values, whatever the module dtype


def mmfs_case(mods, name, *, cfg, B, Lq, n, mask, seed, ref_kind="centre", grid_hw=None,
              dtype=torch.float64, padding=False):
    gen = torch.Generator().manual_seed(seed)
    m = mods.MMFS(**cfg).to(dtype)
    randomise(m, gen)
    shapes1 = [(s, s) for s in cfg["spatial_shapes"]]
    sh, start = level_tables(shapes1 * n)
    hw = sum(h * w for h, w in shapes1)
    query = torch.randn(B, Lq, cfg["d_query"], generator=gen).to(dtype).requires_grad_(True)
    feat = torch.randn(B, n, hw, cfg["d_value"], generator=gen).to(dtype).requires_grad_(True)
    pad = None
    if padding:
        # input_padding_mask [B, n, hw] (mmfs.py:165-172: the projected value rows under it are zeroed): ragged tails of
        # every image's token axis plus scattered pixels
        pad = torch.rand(B, n, hw, generator=gen) < 0.15
        pad[:, :, -5:] = True
        pad[0, 0] = False
    if ref_kind == "centre":
        ref = torch.full((1, Lq, 1, 2), 0.5, dtype=torch.float32)
    elif ref_kind == "boxes":
        # 4-D reference points (cx, cy, w, h), one box per (sample, query, level) (mmfs.py:251-258; no caller of the
        # reference takes this branch -- API fidelity)
        L = len(shapes1) * n
        ctr = torch.rand(B, Lq, L, 2, generator=gen) * 0.6 + 0.2
        wh = torch.rand(B, Lq, L, 2, generator=gen) * 0.5 + 0.1
        ref = torch.cat((ctr, wh), -1).float()
    else:
        gh, gw = grid_hw
        ys = (torch.arange(gh, dtype=torch.float32) + 0.5) / gh
        xs = (torch.arange(gw, dtype=torch.float32) + 0.5) / gw
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        ref = torch.stack((xx.reshape(-1), yy.reshape(-1)), -1)[None, :, None]
        assert ref.shape[1] == Lq
    out = m(query, ref.to(dtype), feat, sh, start, pad, mask)
    g = torch.randn(out.shape, generator=gen).to(dtype)
    out.backward(g)
    arrays = dict(query=query, feat=feat, reference_points=ref, spatial_shapes=sh, level_start_index=start,
                  attention_mask=mask, grad_out=g, out=out, grad_query=query.grad, grad_feat=feat.grad)
    for k, v in m.state_dict().items():
        arrays["param." + k] = v
    for k, p in m.named_parameters():
        if p.grad is not None:
            arrays["grad." + k] = p.grad
    arrays["cfg"] = np.array(repr(cfg))
    if pad is not None:
        arrays["input_padding_mask"] = pad
    save(name, **arrays)


def make_mmfs_goldens(mods, only=None):
    import contextlib, io
    llm = dict(layer_idx=0, d_model=64, d_query=64, d_value=32, d_out=64, n_levels=3, n_heads=4,
               n_points=2, ratio=0.5, offset_init_magnitude=3.0, spatial_shapes=[8, 4, 2],
               base_spatial_shape=4, max_num_image_per_seq=8)
    sd = dict(layer_idx=3, d_model=32, d_query=24, d_value=32, d_out=24, n_levels=4, n_heads=4,
              n_points=3, ratio=1.0, offset_init_magnitude=1, spatial_shapes=[8, 4, 2, 1],
              base_spatial_shape=2, max_num_image_per_seq=6)
    llm8 = dict(llm, n_points=8, n_heads=4, base_spatial_shape=16)
    sd8 = dict(layer_idx=5, d_model=64, d_query=40, d_value=24, d_out=40, n_levels=4, n_heads=4,
               n_points=8, ratio=1.0, offset_init_magnitude=1, spatial_shapes=[8, 4, 2, 1],
               base_spatial_shape=4, max_num_image_per_seq=10)
    mask3 = torch.tensor([[[0, 0, 0], [1, 0, 0], [1, 1, 0], [1, 1, 1], [1, 1, 1]],
                          [[0, 0, 0], [0, 0, 0], [0, 1, 0], [0, 1, 1], [1, 1, 1]]], dtype=torch.float32)
    with contextlib.redirect_stdout(io.StringIO()):
        # ---- round 6: the two API-fidelity branches of MMFS.forward no caller takes (VERDICT r5 missing 3): box
        # reference points (mmfs.py:251-258) and input_padding_mask (mmfs.py:165-172), alone and together
        mmfs_case(mods, "mmfs_p8_llm_boxes", cfg=llm8, B=2, Lq=5, n=3, mask=mask3, seed=60, ref_kind="boxes")
        mmfs_case(mods, "mmfs_p8_sd_padded", cfg=sd8, B=3, Lq=12, n=2,
                  mask=torch.tensor([[1, 1], [0, 0], [0, 1]], dtype=torch.long), seed=61,
                  ref_kind="grid", grid_hw=(4, 3), padding=True)
        mmfs_case(mods, "mmfs_p8_llm_boxes_padded_f32", cfg=llm8, B=2, Lq=5, n=3, mask=mask3, seed=62, ref_kind="boxes",
                  padding=True, dtype=torch.float32)
    if only == "r06":
        return
    with contextlib.redirect_stdout(io.StringIO()):
        # LLM flavour, 3-D float mask [B, Lq, n] with an all-masked row and a partially visible one
        mask3 = torch.tensor([[[0, 0, 0], [1, 0, 0], [1, 1, 0], [1, 1, 1], [1, 1, 1]],
                              [[0, 0, 0], [0, 0, 0], [0, 1, 0], [0, 1, 1], [1, 1, 1]]], dtype=torch.float32)
        mmfs_case(mods, "mmfs_llm_mask3d", cfg=llm, B=2, Lq=5, n=3, mask=mask3, seed=10)
        # decode step: Lq=1 while the mask still has the full history -> last row is used (mmfs.py:161-162)
        mmfs_case(mods, "mmfs_llm_decode", cfg=llm, B=2, Lq=1, n=3, mask=mask3, seed=11)
        # single image, 2-D long mask
        mmfs_case(mods, "mmfs_llm_n1", cfg=llm, B=2, Lq=4, n=1,
                  mask=torch.ones(2, 1, dtype=torch.long), seed=12)
        # SD flavour: per-pixel reference grid, 2-D long mask with a fully masked sample
        mask2 = torch.tensor([[1, 1, 0, 1], [0, 0, 0, 0], [0, 1, 1, 1]], dtype=torch.long)
        mmfs_case(mods, "mmfs_sd_mask2d", cfg=sd, B=3, Lq=12, n=4, mask=mask2, seed=13,
                  ref_kind="grid", grid_hw=(3, 4))
        # fp32 run of the LLM case (what an fp32 implementation should reproduce to ~1e-5)
        mmfs_case(mods, "mmfs_llm_mask3d_f32", cfg=llm, B=2, Lq=5, n=3, mask=mask3, seed=10,
                  dtype=torch.float32)
        # ---- the point counts the decoders really use (modeling_llama_mmfs.py:326-339, sd_mmfs.py:50-53:
        # n_points = 8) and the north star's 4: these reach the fused sampling-plan kernel of the build
        llm8 = dict(llm, n_points=8, n_heads=4, base_spatial_shape=16)
        llm4 = dict(llm, n_points=4, n_heads=8, d_model=128, d_out=48)
        sd8 = dict(layer_idx=5, d_model=64, d_query=40, d_value=24, d_out=40, n_levels=4, n_heads=4,
                   n_points=8, ratio=1.0, offset_init_magnitude=1, spatial_shapes=[8, 4, 2, 1],
                   base_spatial_shape=4, max_num_image_per_seq=10)
        mask4 = torch.tensor([[[0, 0, 0, 0], [1, 0, 0, 0], [1, 1, 0, 0], [1, 1, 1, 0], [1, 1, 1, 1], [0, 1, 0, 1]],
                              [[0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 1, 0], [1, 0, 1, 1], [1, 1, 1, 1], [1, 1, 1, 1]]],
                             dtype=torch.float32)
        mmfs_case(mods, "mmfs_p8_llm_n3", cfg=llm8, B=2, Lq=5, n=3, mask=mask3, seed=40)
        mmfs_case(mods, "mmfs_p8_llm_n4", cfg=llm8, B=2, Lq=6, n=4, mask=mask4, seed=41)
        mmfs_case(mods, "mmfs_p8_llm_n1", cfg=llm8, B=3, Lq=7, n=1, mask=torch.ones(3, 1, dtype=torch.long), seed=42)
        mmfs_case(mods, "mmfs_p8_llm_decode", cfg=llm8, B=2, Lq=1, n=4, mask=mask4, seed=43)
        mmfs_case(mods, "mmfs_p4_llm_n3", cfg=llm4, B=2, Lq=5, n=3, mask=mask3, seed=44)
        mmfs_case(mods, "mmfs_p8_sd_grid", cfg=sd8, B=3, Lq=12, n=2,
                  mask=torch.tensor([[1, 1], [0, 0], [0, 1]], dtype=torch.long), seed=45,
                  ref_kind="grid", grid_hw=(4, 3))
        mmfs_case(mods, "mmfs_p8_sd_n1", cfg=sd8, B=2, Lq=16, n=1, mask=torch.ones(2, 1, dtype=torch.long),
                  seed=46, ref_kind="grid", grid_hw=(4, 4))
        mmfs_case(mods, "mmfs_p8_llm_n3_f32", cfg=llm8, B=2, Lq=5, n=3, mask=mask3, seed=40, dtype=torch.float32)
        mmfs_case(mods, "mmfs_p8_sd_grid_f32", cfg=sd8, B=3, Lq=12, n=2,
                  mask=torch.tensor([[1, 1], [0, 0], [0, 1]], dtype=torch.long), seed=45,
                  ref_kind="grid", grid_hw=(4, 3), dtype=torch.float32)


# ----------------------------------------------------------------------------- G5: blocks
def make_block_goldens():
    import contextlib, io
    from transformers import LlamaConfig
    sd, llama = import_reference_blocks()
    gen = torch.Generator().manual_seed(20)
    dt = torch.float64
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = LlamaConfig(hidden_size=64, num_attention_heads=4, intermediate_size=128,
                          num_hidden_layers=1, rms_norm_eps=1e-6, max_position_embeddings=64)
        cfg.image_embed_dim = 32
        cfg.cross_attention_frequency = 4
        cfg.spatial_shapes = [8, 4, 2]
        att = llama.LlamaMMFSAttention(cfg, layer_idx=0).to(dt)
        randomise(att, gen)
        with torch.no_grad():
            att.gate.fill_(0.7)
            att.norm1.weight.copy_(1 + 0.1 * torch.randn(64, generator=gen))
            att.norm2.weight.copy_(1 + 0.1 * torch.randn(32, generator=gen))
    B, Lq, n = 2, 6, 2
    hidden = torch.randn(B, Lq, 64, generator=gen).to(dt).requires_grad_(True)
    feats = torch.randn(B, n, 64 + 16 + 4, 32, generator=gen).to(dt).requires_grad_(True)
    mask = torch.tensor([[[0, 0], [1, 0], [1, 0], [1, 1], [1, 1], [1, 1]],
                         [[1, 0], [1, 0], [1, 1], [1, 1], [1, 1], [1, 1]]], dtype=torch.float32)
    out = att(hidden, feats, mask)
    g = torch.randn(out.shape, generator=gen).to(dt)
    out.backward(g)
    arrays = dict(hidden=hidden, feats=feats, mask=mask, out=out, grad_out=g,
                  grad_hidden=hidden.grad, grad_feats=feats.grad)
    arrays.update({"param." + k: v for k, v in att.state_dict().items()})
    arrays.update({"grad." + k: p.grad for k, p in att.named_parameters() if p.grad is not None})
    save("block_llama_mmfs_attention", **arrays)

    # MMFSBlock (sd_mmfs.py:44-145): tiny config
    with contextlib.redirect_stdout(io.StringIO()):
        blk = sd.MMFSBlock(attn_dim=32, query_dim=16, feat_dim=32, num_heads=4, n_points=2, n_levels=3,
                           gradient_checkpointing=False, grid_size=8, spatial_shapes=[8, 4, 2],
                           base_spatial_shape=4, max_num_image_per_seq=5).to(dt)
        randomise(blk, gen)
        with torch.no_grad():
            blk.pos_embed.copy_(torch.from_numpy(
                importlib.import_module("mm_interleaved.models.utils.pos_embed")
                .get_2d_sincos_pos_embed(16, 8, cls_token=False)).to(dt))
    B, n = 2, 2
    sample = torch.randn(B, 16, 4, 4, generator=gen).to(dt).requires_grad_(True)
    ms_feat = torch.randn(B, n, 84, 32, generator=gen).to(dt).requires_grad_(True)
    ms_mask = torch.tensor([[1, 1], [1, 0]], dtype=torch.long)
    out = blk(sample, ms_feat, ms_mask, [(8, 8), (4, 4), (2, 2)])
    g = torch.randn(out.shape, generator=gen).to(dt)
    out.backward(g)
    arrays = dict(sample=sample, ms_feat=ms_feat, ms_mask=ms_mask, out=out, grad_out=g,
                  grad_sample=sample.grad, grad_ms_feat=ms_feat.grad)
    arrays.update({"param." + k: v for k, v in blk.state_dict().items()})
    arrays.update({"grad." + k: p.grad for k, p in blk.named_parameters() if p.grad is not None})
    save("block_sd_mmfs_block", **arrays)

    # MMFSNet (sd_mmfs.py:154-272): 2 resolution stages -> 1 + 2 + 1 + 2 = 6 down blocks + mid
    with contextlib.redirect_stdout(io.StringIO()):
        net = sd.MMFSNet(input_channel=32, block_out_channels=[16, 24], layers_per_block=2,
                         downsample_factor=8, n_levels=3, n_points=2, gradient_checkpointing=False,
                         spatial_shapes=[64, 32, 16]).to(dt)
        randomise(net, gen)
    B, n = 1, 2
    res_shapes = [(16, 8), (16, 8), (16, 8), (16, 4), (24, 4), (24, 4)]
    res = [torch.randn(B, c, s, s, generator=gen).to(dt) for c, s in res_shapes]
    mid = torch.randn(B, 24, 4, 4, generator=gen).to(dt)
    feats = [torch.randn(B, n, 32, s, s, generator=gen).to(dt) for s in (8, 4, 2)]
    ms_mask = torch.tensor([[1, 1]], dtype=torch.long)
    new_mid, new_res = net(mid, res, feats, ms_mask)
    arrays = dict(mid=mid, ms_mask=ms_mask, new_mid=new_mid)
    arrays.update({f"res.{i}": r for i, r in enumerate(res)})
    arrays.update({f"new_res.{i}": r for i, r in enumerate(new_res)})
    arrays.update({f"feat.{i}": r for i, r in enumerate(feats)})
    arrays.update({"param." + k: v for k, v in net.state_dict().items()})
    # ... and a TRAINING step of the reference (round 5): gradient checkpointing on in every block
    # (sd_mmfs.py:138-141), cotangents for all seven outputs -> the gradient of every input and of EVERY parameter.  (Drawn
    # after everything above: the arrays of the forward fixture keep their values.)
    for blk in list(net.mmfs_down_blocks) + [net.mmfs_mid_block]:
        blk.gradient_checkpointing = True
    net.train()
    res_t = [r.clone().requires_grad_(True) for r in res]
    mid_t = mid.clone().requires_grad_(True)
    feats_t = [f.clone().requires_grad_(True) for f in feats]
    out_mid, out_res = net(mid_t, res_t, feats_t, ms_mask)
    cot_mid = torch.randn(out_mid.shape, generator=gen).to(dt)
    cot_res = [torch.randn(r.shape, generator=gen).to(dt) for r in out_res]
    loss = (out_mid * cot_mid).sum()
    for r, c in zip(out_res, cot_res):
        loss = loss + (r * c).sum()
    loss.backward()
    assert float((out_mid - new_mid).abs().max()) == 0.0
    arrays.update({"train.cot_mid": cot_mid, "train.grad_mid": mid_t.grad})
    arrays.update({f"train.cot_res.{i}": c for i, c in enumerate(cot_res)})
    arrays.update({f"train.grad_res.{i}": r.grad for i, r in enumerate(res_t)})
    arrays.update({f"train.grad_feat.{i}": f.grad for i, f in enumerate(feats_t)})
    arrays.update({"train.grad." + k: p.grad for k, p in net.named_parameters() if p.grad is not None})
    save("block_sd_mmfs_net", **arrays)


# ----------------------------------------------------------------------------- G7: encoder MSDeformAttn
def import_reference_encoder_ops():
    """encoders/vit_adapter/ops as its own package (a second package named ``ops`` in the reference)."""
    name = "ref_vit_adapter_ops"
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, "mm_interleaved/models/encoders/vit_adapter/ops")]
        sys.modules[name] = m
    funcs = importlib.import_module(name + ".functions.ms_deform_attn_func")
    mod = importlib.import_module(name + ".modules.ms_deform_attn")

    class CoreAsFunction:
        @staticmethod
        def apply(value, shapes, start, loc, attn, im2col_step):
            return funcs.ms_deform_attn_core_pytorch(value, shapes, loc, attn)

    mod.MSDeformAttnFunction = CoreAsFunction
    return mod


def encoder_case(mod, name, *, d_model, n_levels, n_heads, n_points, ratio, shapes, B, Lq, seed, ref_dim=2,
                 dtype=torch.float64, padding=False):
    """encoders/vit_adapter/ops/modules/ms_deform_attn.py:28-131 as the ViT-Adapter calls it
    (adapter_modules.py:108-154: injector = ViT tokens query the 3-level pyramid, extractor = pyramid
    tokens query the single ViT map), parameters randomised (the module starts with zero weights)."""
    import contextlib, io
    gen = torch.Generator().manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = mod.MSDeformAttn(d_model=d_model, n_levels=n_levels, n_heads=n_heads, n_points=n_points, ratio=ratio).to(dtype)
    randomise(m, gen)
    sh, start = level_tables(shapes)
    S = int(sh.prod(1).sum())
    query = torch.randn(B, Lq, d_model, generator=gen).to(dtype).requires_grad_(True)
    feat = torch.randn(B, S, d_model, generator=gen).to(dtype).requires_grad_(True)
    ref = torch.rand(B, Lq, n_levels, ref_dim, generator=gen)
    if ref_dim == 4:
        ref[..., 2:] = ref[..., 2:] * 0.3 + 0.05
    pad = None
    if padding:
        pad = torch.rand(B, S, generator=gen) < 0.2
    out = m(query, ref.to(dtype), feat, sh, start, pad)
    g = torch.randn(out.shape, generator=gen).to(dtype)
    out.backward(g)
    arrays = dict(query=query, feat=feat, reference_points=ref, spatial_shapes=sh, level_start_index=start,
                  grad_out=g, out=out, grad_query=query.grad, grad_feat=feat.grad,
                  cfg=np.array(repr(dict(d_model=d_model, n_levels=n_levels, n_heads=n_heads, n_points=n_points, ratio=ratio))))
    if pad is not None:
        arrays["padding_mask"] = pad
    arrays.update({"param." + k: v for k, v in m.state_dict().items()})
    arrays.update({"grad." + k: p.grad for k, p in m.named_parameters() if p.grad is not None})
    save(name, **arrays)


def make_encoder_goldens():
    mod = import_reference_encoder_ops()
    # geometry of the adapter (vit_adapter_hf.py:58-65: d_model 1024, 16 heads, deform_ratio 0.5 -> D = 32, P = 4)
    # at reduced width: 4 heads, the same D = 32 (the head width picks the kernel, not the head count)
    inj = dict(d_model=256, n_levels=3, n_heads=4, n_points=4, ratio=0.5, shapes=[(8, 8), (4, 4), (2, 2)])
    ext = dict(d_model=256, n_levels=1, n_heads=4, n_points=4, ratio=0.5, shapes=[(4, 4)])
    encoder_case(mod, "enc_injector", **inj, B=2, Lq=16, seed=50)
    encoder_case(mod, "enc_extractor", **ext, B=2, Lq=84, seed=51)
    # the two branches no caller of the reference takes, kept for API fidelity: box reference points, padding mask
    encoder_case(mod, "enc_boxes_padded", d_model=64, n_levels=2, n_heads=4, n_points=2, ratio=1.0,
                 shapes=[(5, 3), (2, 4)], B=2, Lq=7, seed=52, ref_dim=4, padding=True)


# ----------------------------------------------------------------------------- G6: bank builders
def make_bank_goldens():
    fns = load_reference_methods(os.path.join(REF, "mm_interleaved/models/mm_interleaved.py"),
                                 {"_prepare_mmfs_features_for_mm_decoder",
                                  "_prepare_mmfs_features_for_image_decoder"})
    BOS, EOS, PAD, SOI, IMG = 1, 2, 31999, 32000, 32001
    self = types.SimpleNamespace(
        special_token_dict=dict(bos_token_id=BOS, eos_token_id=EOS, pad_token_id=PAD,
                                soi_token_id=SOI, image_token_id=IMG),
        spatial_shapes=[8, 4, 2])
    gen = torch.Generator().manual_seed(30)
    T = lambda n: torch.randint(5, 100, (n,), generator=gen).tolist()
    seqs = [
        [BOS] + T(2) + [SOI, IMG, IMG] + T(3) + [SOI, IMG, IMG] + T(2) + [EOS, BOS] + T(1) + [SOI, IMG, IMG] + T(2),
        [BOS] + T(5) + [SOI, IMG, IMG] + T(8) + [EOS] + [PAD] * 7,
        [BOS] + T(1) + [EOS, BOS, SOI, IMG, IMG] + T(2) + [SOI, IMG, IMG] + T(10) + [EOS],
    ]
    Lmax = max(len(s) for s in seqs)
    text_ids = torch.tensor([s + [PAD] * (Lmax - len(s)) for s in seqs], dtype=torch.long)
    num_image_per_seq = torch.tensor([3, 1, 2])
    n_img = int(num_image_per_seq.sum())
    ms = [torch.randn(n_img, 6, s, s, generator=gen) for s in (16, 8, 4, 2)]   # 16 is filtered out
    out = fns["_prepare_mmfs_features_for_mm_decoder"](self, text_ids, num_image_per_seq, ms)
    arrays = dict(text_ids=text_ids, num_image_per_seq=num_image_per_seq,
                  cross_attention_mask=out["cross_attention_mask"], mmfs_features_mm=out["mmfs_features_mm"])
    arrays.update({f"ms.{i}": f for i, f in enumerate(ms)})
    # image decoder bank: previous image of the same document only
    nearest_bos = []
    for b, s in enumerate(seqs):
        last = 0
        for t, tok in enumerate(s):
            if tok == BOS:
                last = t
            if tok == SOI:
                nearest_bos.append(last)
    nearest_bos = torch.tensor(nearest_bos)
    feats_i, mask_i = fns["_prepare_mmfs_features_for_image_decoder"](
        self, ms[1:], text_ids, nearest_bos, num_image_per_seq)
    arrays["nearest_bos_idxs"] = nearest_bos
    arrays["img_mask"] = mask_i
    arrays.update({f"img_feat.{i}": f for i, f in enumerate(feats_i)})
    save("bank_builders", **arrays)


if __name__ == "__main__":
    funcs, mods = import_reference_ops()
    if len(sys.argv) > 1 and sys.argv[1] == "r06":          # only the fixtures round 6 added (the others are unchanged)
        make_mmfs_goldens(mods, only="r06")
        sys.exit(0)
    make_op_goldens(funcs.ms_deform_attn_core_pytorch)
    make_mmfs_goldens(mods)
    make_bank_goldens()
    make_block_goldens()
    make_encoder_goldens()
