"""UNet-side feature synchronizer blocks.

``MMFSBlock`` and ``MMFSNet`` mirror mm_interleaved/models/decoders/sd_mmfs.py:44-272:
LayerNorm(query) + bicubically resized sin-cos position embedding, LayerNorm(features),
MMFS with one reference point per UNet pixel, zero-initialised 1x1 convolution; the net
applies one block to each of the 12 down-path residuals and one to the mid sample,
additively.  State-dict keys are the reference's (``query_norm.*``, ``feat_norm.*``,
``mmfs.*``, ``pos_embed``, ``conv.*``; ``mmfs_down_blocks.{i}.*``, ``mmfs_mid_block.*``).

The position-embedding helpers restate mm_interleaved/models/utils/pos_embed.py:16-97
(constant tables).
"""
import math
from functools import partial

import torch
import torch.nn.functional as F
import torch.utils.checkpoint as cp
from torch import nn

from ..bank import gather_bank
from .. import graphed as _graphed
from ..graphed import graphed_call, memory_is_plentiful
from ..functions.linear_func import _split_k, token_linear
from ..functions.query_func import QueryPrepFunction, TokensAddFunction, layout_supported, query_prep, tokens_add
from ..levels import CacheInvalidation, cache_epoch, hook_free, make_level_tables, tensor_version
from ..modules.mmfs import MMFS, FoldedLinear


# ------------------------------------------------------------------ constant tables
def sincos_1d(dim, pos):
    """[M] positions -> [M, dim]: sin | cos halves, frequencies 1/10000^(2i/dim) (pos_embed.py:78-97)."""
    assert dim % 2 == 0
    omega = 1.0 / (10000.0 ** (torch.arange(dim // 2, dtype=torch.float32) / (dim / 2.0)))
    ang = pos.reshape(-1).to(torch.float32)[:, None] * omega[None, :]
    return torch.cat((ang.sin(), ang.cos()), 1)


def sincos_2d(dim, grid_size):
    """[grid*grid, dim] table, row-major over (y, x): first half encodes y, second half x
    (pos_embed.py:47-75)."""
    assert dim % 2 == 0
    ys, xs = torch.meshgrid(torch.arange(grid_size, dtype=torch.float32),
                            torch.arange(grid_size, dtype=torch.float32), indexing="ij")
    return torch.cat((sincos_1d(dim // 2, ys), sincos_1d(dim // 2, xs)), 1)


def resize_pos_embed(table, n_tokens):
    """[g*g, C] -> [n_tokens, C] by bicubic interpolation when the square sizes differ
    (pos_embed.py:16-40, without its cls-token branch: MMFS tables have none)."""
    src = int(math.sqrt(table.size(0)))
    tgt = int(math.sqrt(n_tokens))
    if src == tgt:
        return table
    out = F.interpolate(table.float().reshape(1, src, src, -1).permute(0, 3, 1, 2), size=(tgt, tgt),
                        mode="bicubic", align_corners=False)
    return out.permute(0, 2, 3, 1).flatten(0, 2).to(table.dtype)


_ref_cache = {}


def pixel_reference_points(h, w, device):
    """[1, h*w, 1, 2] pixel centres (x, y) of an h x w map, normalised (sd_mmfs.py:15-28)."""
    key = (h, w, str(device))
    hit = _ref_cache.get(key)
    if hit is None:
        # (cached across calls: never an inference tensor, see mmfs_amd.levels.make_level_tables)
        with torch.inference_mode(False), torch.no_grad():
            ys = (torch.arange(h, dtype=torch.float32, device=device) + 0.5) / h
            xs = (torch.arange(w, dtype=torch.float32, device=device) + 0.5) / w
            yy, xx = torch.meshgrid(ys, xs, indexing="ij")
            hit = torch.stack((xx.reshape(-1), yy.reshape(-1)), -1)[None, :, None, :].contiguous()
        _ref_cache[key] = hit
    return hit


def deform_inputs(sample, spatial_shapes=((8, 8),), n_images=1):
    """(reference_points, spatial_shapes, level_start_index) for a [B, C, h, w] UNet sample
    (sd_mmfs.py:31-41); ``spatial_shapes`` are one image's levels."""
    _, _, h, w = sample.shape
    shapes, start, _ = make_level_tables(spatial_shapes, n_images, sample.device)
    return pixel_reference_points(h, w, sample.device), shapes, start


class _ProjectAll(torch.autograd.Function):
    """``x [T, C], wt [K, C, D], bias [K, D]  ->  K tensors x @ wt[k] + bias[k]`` (slices of ONE [K, T, D] buffer, one batched
    GEMM).  Written as a Function because of its backward: left to autograd, every consumer's gradient of "its" slice of
    the stacked result becomes a zero-filled [K, T, D] tensor with one slice set, and the K of them are added up -- at the
    image decoder's shape 13 fills and 13 adds of 1.16 GB each, 10.8 ms of a 37 ms training step (r04t).  Here the K
    incoming gradients are taken as they come: every weight gradient x^T g_k is a product whose sum over the T tokens is
    cut into chunks (``linear_func._split_k``: a [C, T] x [T, D] product is 16 output tiles for 256 CUs, each walking
    all 43 520 tokens -- 264 us apiece, and as ONE batched GEMM over the K gradients copied side by side 1.93 ms + 0.57 ms
    of copies, r04zk), the input gradient K GEMMs accumulating in place."""

    @staticmethod
    def forward(ctx, x, wt, bias):
        y = torch.matmul(x[None], wt)
        y += bias[:, None, :]
        ctx.save_for_backward(x, wt)
        return tuple(y.unbind(0))

    @staticmethod
    def backward(ctx, *grads):
        x, wt = ctx.saved_tensors
        K, T, C, D = wt.shape[0], x.shape[0], wt.shape[1], wt.shape[2]
        gs = [None if g is None else g.reshape(T, D).contiguous() for g in grads]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            for k, g in enumerate(gs):
                if g is not None:
                    gx = torch.mm(g, wt[k].t()) if gx is None else gx.addmm_(g, wt[k].t())
            if gx is None:
                gx = torch.zeros_like(x)
        if ctx.needs_input_grad[1]:
            S = _split_k(T, C, D) if (x.is_cuda and x.is_contiguous()) else 1
            gw = torch.empty_like(wt)                                      # [K, C, D]
            for k, g in enumerate(gs):
                if g is None:
                    gw[k].zero_()
                elif S > 1:
                    torch.sum(torch.bmm(x.view(S, T // S, C).transpose(1, 2), g.view(S, T // S, D)), 0, out=gw[k])
                else:
                    torch.mm(x.t(), g, out=gw[k])
        if ctx.needs_input_grad[2]:
            gb = torch.stack([torch.zeros(D, dtype=wt.dtype, device=wt.device) if g is None else g.sum(0) for g in gs])
        return gx, gw, gb


# ------------------------------------------------------------------ blocks
class MMFSBlock(CacheInvalidation, nn.Module):
    layout_kernels_in_training = True
    fold_conv = True
    # a checkpointed call as two HIP graphs -- the forward, and "forward again + backward" -- once its shapes have been
    # seen a few times (mmfs_amd/graphed.py): the eager training step of the unchanged trainer is bound by the host
    # otherwise (4 075 launches per step at BASELINE config 4).  False: ``torch.utils.checkpoint`` always.
    graph_checkpoints = True
    # ... and when the device has memory to spare, the recorded forward KEEPS its activations (in the graph's own pool) and
    # the second graph is the backward alone: no recomputation.  Checkpointing exists to save memory -- the 13 blocks'
    # activations at BASELINE config 4 are ~2 GB of a 288 GB device -- and costs a third of the step's kernels.  "auto":
    # while at least half of the device's memory is free (``mmfs_amd.graphed.memory_is_plentiful``); False: the recorded
    # call recomputes like the checkpoint it replaces; True: always keeps.  Same gradients bit for bit either way.
    graph_keeps_activations = "auto"
    _behaviour_flags = ("layout_kernels_in_training", "fold_conv", "gradient_checkpointing")      # (part of a recorded call's key)

    def __init__(self, attn_dim=1024, query_dim=320, feat_dim=1024, num_heads=16, n_points=8,
                 n_levels=1, deform_ratio=1.0, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                 gradient_checkpointing=False, grid_size=64, offset_init_magnitude=1,
                 max_num_image_per_seq=10, spatial_shapes=[16], base_spatial_shape=8, layer_idx=0):
        super().__init__()
        self.gradient_checkpointing = gradient_checkpointing
        self.query_norm = norm_layer(query_dim)
        self.feat_norm = norm_layer(feat_dim)
        self.mmfs = MMFS(d_model=attn_dim, d_query=query_dim, d_value=feat_dim, d_out=query_dim,
                         n_levels=n_levels, n_heads=num_heads, n_points=n_points, ratio=deform_ratio,
                         offset_init_magnitude=offset_init_magnitude, spatial_shapes=spatial_shapes,
                         base_spatial_shape=base_spatial_shape,
                         max_num_image_per_seq=max_num_image_per_seq, layer_idx=layer_idx)
        self.pos_embed = nn.Parameter(sincos_2d(query_dim, grid_size), requires_grad=False)
        self.conv = nn.Conv2d(query_dim, query_dim, kernel_size=1, stride=1)
        nn.init.zeros_(self.conv.weight)          # zero_module (sd_mmfs.py:88-94, 148-151)
        nn.init.zeros_(self.conv.bias)
        self._conv_fold = FoldedLinear()

    def _reset_parameters(self):
        self.mmfs._reset_parameters()

    def _pos_table(self, n_tokens):
        """The position table at this block's resolution.  The reference interpolates it on every
        call (sd_mmfs.py:127-131); the table is a frozen parameter, so the result only changes when
        the parameter does -- and the framework's bicubic kernel on a [1, C, 64, 64] map is slow
        (2.4 ms per call measured on MI355X, 80 % of a whole MMFSNet forward): keep it."""
        pe = self.pos_embed
        if pe.requires_grad:                      # someone is training it: stay in the graph
            return resize_pos_embed(pe, n_tokens)
        key = (cache_epoch(), n_tokens, pe.data_ptr(), tensor_version(pe), pe.dtype, pe.device)
        hit = self.__dict__.get("_pos_cache")
        if hit is None or hit[0] != key:
            # (kept across calls: a table made during an inference_mode() pass must still be usable
            # by a later training step, so it is computed outside the mode on a plain copy)
            with torch.inference_mode(False), torch.no_grad():
                src = pe.detach().clone() if pe.is_inference() else pe.detach()
                hit = (key, resize_pos_embed(src, n_tokens).clone())
            self.__dict__["_pos_cache"] = hit
        return hit[1]

    def _conv_is_pointwise(self):
        """The block's convolution as the reference builds it (sd_mmfs.py:88-94: 1x1, stride 1, dense): a per-token
        linear map.  Anything else a caller put there is called as the convolution it is."""
        c = self.conv
        return (type(c) is nn.Conv2d and c.kernel_size == (1, 1) and c.stride == (1, 1) and c.padding == (0, 0)
                and c.dilation == (1, 1) and c.groups == 1 and c.padding_mode == "zeros"
                and hook_free(c))

    def _layout_kernels(self, sample):
        """Whether the two layout changes around the block run as one kernel each (csrc/mmfs_query.hip): a plain affine
        LayerNorm over the channels, one 16-bit storage type throughout, no autocast.  (With gradients the same kernels
        run inside ``QueryPrepFunction`` / ``TokensAddFunction``; ``layout_kernels_in_training = False`` keeps a training
        step on the framework's kernels.)"""
        n = self.query_norm
        if torch.is_grad_enabled() and not self.layout_kernels_in_training:
            return False
        return (not torch.is_autocast_enabled() and type(n) is nn.LayerNorm and n.elementwise_affine and n.bias is not None
                and tuple(n.normalized_shape) == (sample.shape[1],) and n.weight.dtype == sample.dtype
                and n.bias.dtype == sample.dtype and self.pos_embed.dtype == sample.dtype
                and self._conv_is_pointwise() and layout_supported(sample))

    def _inner(self, sample, ms_feat, ms_feat_mask, spatial_shapes, value=None, image_ranks=None, residual=None, normed=None):
        B, C, H, W = sample.shape
        if value is None and normed is not None:
            # the bank already normalised WITHOUT the affine (MMFSNet, shared by its blocks): this block's affine folds
            # into its projection, value_proj(g * xhat + b) = (W diag g) xhat + (W b + bias)
            proj, ln = self.mmfs.value_proj, self.feat_norm
            value = F.linear(normed, proj.weight * ln.weight, F.linear(ln.bias, proj.weight, proj.bias))
        n_images = ms_feat_mask.shape[-1]
        ref, shapes, start = deform_inputs(sample, spatial_shapes, n_images)
        fast = self._layout_kernels(sample)
        if fast:
            # "b c h w -> b (h w) c", the normalisation and the position term in one pass over the residual
            pos = self._pos_table(H * W)
            if torch.is_grad_enabled():
                query = QueryPrepFunction.apply(sample, self.query_norm.weight, self.query_norm.bias, self.query_norm.eps, pos)
            else:
                query = query_prep(sample, self.query_norm.weight, self.query_norm.bias, self.query_norm.eps, pos)
        else:
            query = self.query_norm(sample.flatten(2).transpose(1, 2))            # b (h w) c
            query = query + self._pos_table(H * W)
        # the zero-initialised 1x1 convolution follows the output projection with nothing non-linear between them: without
        # gradients the two are ONE GEMM on kept product weights (FoldedLinear; ``fold_conv = False``: two)
        # (not under autocast: the two layers then round to the autocast type between them, as the reference's do)
        proj = self.mmfs.output_proj
        one_gemm = (self.fold_conv and not self.training and not torch.is_grad_enabled() and not torch.is_autocast_enabled()
                    and self._conv_is_pointwise() and type(proj) is nn.Linear and hook_free(proj))
        folded = None
        if one_gemm:
            folded = self._conv_fold.get(proj.weight, proj.bias, self.conv.weight.view(C, C), self.conv.bias)
        out = self.mmfs(query, ref, self.feat_norm(ms_feat) if value is None else ms_feat, shapes, start,
                        input_padding_mask=None, attention_mask=ms_feat_mask, value=value, image_ranks=image_ranks,
                        output_weights=folded)
        # the zero-initialised 1x1 convolution (sd_mmfs.py:88-94, 146) is a per-token linear map:
        # applied on the token-major tensor it is one GEMM each way (the convolution library's 1x1
        # backward took 0.45 ms per block at B=8, the GEMMs take ~0.05)
        if self._conv_is_pointwise():
            if not one_gemm:
                out = token_linear(out, self.conv.weight.view(C, C), self.conv.bias) if hook_free(self.conv) \
                    else self.conv(out.transpose(1, 2).reshape(B, C, H, W)).flatten(2).transpose(1, 2)
            if fast and residual is not None and residual.shape == sample.shape and residual.dtype == out.dtype:
                # "b (h w) c -> b c h w" + the caller's add
                return TokensAddFunction.apply(out, residual) if torch.is_grad_enabled() else tokens_add(out, residual)
            out = out.transpose(1, 2).reshape(B, C, H, W)
        else:
            out = self.conv(out.transpose(1, 2).reshape(B, C, H, W))
        return out if residual is None else residual + out

    def forward(self, sample, ms_feat, ms_feat_mask, spatial_shapes, value=None, image_ranks=None, residual=None, normed=None):
        """sample [B, C_q, H, W]; ms_feat [B, n, sum_l H_l*W_l, C_v]; ms_feat_mask [B, n];
        spatial_shapes: the levels of ONE image, list of (H_l, W_l)  ->  [B, C_q, H, W].
        ``value`` (an addition to sd_mmfs.py:121-146): this block's
        ``mmfs.value_proj(feat_norm(ms_feat))`` when the caller already has it (``MMFSNet``); ``image_ranks``
        (another): ``mmfs._image_relpos(ms_feat_mask, ...)``, a function of the mask only, made once for all blocks;
        ``residual`` (another): a [B, C_q, H, W] tensor the result is added to -- the add ``MMFSNet`` does with every
        block's output (sd_mmfs.py:262-270), here so that it can share the pass that restores the layout; ``normed``
        (another): ``layer_norm(ms_feat)`` without the affine, when the caller has it -- the block then folds its
        ``feat_norm`` affine into its value projection (inside its checkpoint: the projected bank is not kept)."""
        spatial_shapes = [tuple(int(v) for v in s) for s in spatial_shapes]
        if self.gradient_checkpointing and self.training:
            # the op is stateless and re-entrant: the forward is simply re-run in backward
            # (a projected ``value`` is an input of the checkpoint: kept, not recomputed)
            # (``residual`` is an input the caller holds anyway -- for ``MMFSNet`` the sample itself)
            # (no random numbers inside: nothing of the generator's state to save -- which would also be a host round
            # trip that a HIP-graph capture of the step refuses)
            def plain():
                return cp.checkpoint(self._inner, sample, ms_feat, ms_feat_mask, spatial_shapes, value, image_ranks, residual,
                                     normed, use_reentrant=False, preserve_rng_state=False)
            if self.graph_checkpoints:
                keep = self.graph_keeps_activations
                if keep == "auto":
                    # decided ONCE per block and latched (asked on every forward, the answer is part of the key --
                    # free memory crossing the mark mid-run recorded a second set of graphs next to the first, just as
                    # memory became scarce); not asked at all where the call takes the plain path anyway
                    keep = self.__dict__.get("_keeps_latched")
                    if keep is None:
                        if not (sample.is_cuda and _graphed.enabled) or torch.cuda.is_current_stream_capturing():
                            keep = False
                        else:
                            keep = self.__dict__["_keeps_latched"] = bool(memory_is_plentiful(sample.device))
                return graphed_call(self, self._inner, (sample, ms_feat, ms_feat_mask, spatial_shapes, value, image_ranks,
                                                        residual, normed), recompute=not keep, plain=plain)
            return plain()
        return self._inner(sample, ms_feat, ms_feat_mask, spatial_shapes, value, image_ranks, residual, normed)


class ProjectedFeatures:
    """``value_proj(feat_norm(bank))`` of every block of an ``MMFSNet`` for one feature bank:
    ``values[k]`` is block k's [B, n, sum_l H_l*W_l, d_inner] (the mid block last).  Made by
    ``MMFSNet.project_features``; accepted by ``MMFSNet.forward`` in place of the feature list."""

    def __init__(self, values, bank, shapes, sources=None, weights=None):
        self.values, self.bank, self.shapes = values, bank, shapes
        # what it was computed from (identity cache of MMFSNet.forward): the feature tensors with
        # their versions, and (storage, version) of every parameter that went in
        self.sources, self.weights = sources, weights

    def matches(self, feats, weights):
        return (self.sources is not None and len(feats) == len(self.sources) and weights == self.weights
                and all(f is s and tensor_version(f) == v for f, (s, v) in zip(feats, self.sources)))


class MMFSNet(CacheInvalidation, nn.Module):
    # The 13 blocks read the SAME feature bank (sd_mmfs.py:247-270).  With ``fused_schedule`` the
    # bank is normalised once -- LayerNorm statistics do not depend on the block, only the affine
    # does, and that folds into the projection:
    #     value_proj_k(gamma_k * xhat + beta_k) = (W_k diag(gamma_k)) xhat + (W_k beta_k + b_k)
    # -- and each block gets its value projection from the shared xhat: 1 normalisation pass over
    # the [B, n, 5440, 1024] bank instead of 13.  ``cache_projected_features``: outside autograd
    # (sampling: 30 denoising steps x the same feature tensors, sd_pipeline_monkey_patch.py:172-200)
    # the projections of the last bank are kept and reused while the caller passes the same,
    # unmodified feature tensors.  Both are additions; off -> the reference's schedule.
    fused_schedule = True
    cache_projected_features = True
    share_normalised_bank = True        # training under gradient checkpointing: see forward
    # Training under gradient checkpointing, round 4: project the bank for all blocks ONCE, as one batched GEMM, OUTSIDE
    # the checkpoints, and hand every block its projection as a checkpoint input (kept through the step, not recomputed).
    # The reference recomputes LayerNorm + projection inside every checkpoint to save memory -- 13 bank-sized tensors,
    # 1.16 GB at B = 8 in bf16: a quarter of a percent of this GPU's 288 GB -- and pays 26 forward projections and 13
    # weight-gradient GEMMs of the worst shape for it.  False: the round-3 schedule (the reference's
    # memory behaviour); True: always; "auto" (default, round 5 / ADVICE r4): only while the kept projections are a small part
    # of the device memory that is free when the step starts (``project_once_budget`` of it) -- on a smaller part, or with
    # a bank of many images, the step falls back to recomputing inside the checkpoints instead of running out of memory.
    project_once_in_training = "auto"
    project_once_budget = 0.125

    def _project_once(self, mmfs_features):
        """Whether this training step keeps all blocks' projections of the bank (see ``project_once_in_training``)."""
        mode = self.project_once_in_training
        if mode is True or mode is False:
            return mode
        if isinstance(mmfs_features, ProjectedFeatures):
            return True                                    # (the caller made them: they exist already)
        feats = mmfs_features
        try:
            ref = feats[0]
            # (decided once per bank geometry: the driver call that asks for the free memory does not belong in every
            # step, and not inside a stream capture -- mmfs_amd.graphs.GraphedTrainingStep warms up eagerly first)
            key = (tuple(tuple(f.shape) for f in feats), ref.dtype, str(ref.device), self.project_once_budget)
            memo = self.__dict__.get("_project_once_memo")
            if memo is not None and memo[0] == key:
                return memo[1]
            pixels = sum(int(f.shape[-1]) * int(f.shape[-2]) for f in feats)
            n_tok = int(ref.shape[0]) * int(ref.shape[1]) * pixels
            width = max(int(b.mmfs.value_proj.out_features) for b in self._blocks())
            need = len(list(self._blocks())) * n_tok * width * ref.element_size()
            if ref.is_cuda and torch.cuda.is_current_stream_capturing():
                return True                                # (a capture without a warm-up step: the round-4 default)
            free = torch.cuda.mem_get_info(ref.device)[0] if ref.is_cuda else (1 << 62)
            res = need <= self.project_once_budget * free
            self.__dict__["_project_once_memo"] = (key, res)
            return res
        except Exception:
            return False

    def __init__(self, input_channel, block_out_channels, layers_per_block, downsample_factor=1,
                 n_levels=4, n_points=8, gradient_checkpointing=True, spatial_shapes=[64, 32, 16, 8]):
        super().__init__()
        self.downsample_factor = downsample_factor
        sd_shapes = [s // downsample_factor for s in spatial_shapes]

        def block(query_dim, index):
            # three UNet residuals share a resolution: residual i is at sd_shapes[i // 3]
            return MMFSBlock(query_dim=query_dim, feat_dim=input_channel, n_points=n_points,
                             n_levels=n_levels, gradient_checkpointing=gradient_checkpointing,
                             grid_size=64 // downsample_factor, spatial_shapes=spatial_shapes,
                             base_spatial_shape=sd_shapes[index // 3], layer_idx=index)

        # the UNet's down path: conv_in residual, then per stage `layers_per_block` resnet
        # residuals plus one downsampler residual (not after the last stage) (sd_mmfs.py:185-211)
        channels = [block_out_channels[0]]
        for i, ch in enumerate(block_out_channels):
            channels += [ch] * layers_per_block
            if i != len(block_out_channels) - 1:
                channels.append(ch)
        self.mmfs_down_blocks = nn.ModuleList(block(ch, i) for i, ch in enumerate(channels))
        mid = MMFSBlock(query_dim=block_out_channels[-1], feat_dim=input_channel, n_points=n_points,
                        n_levels=n_levels, gradient_checkpointing=gradient_checkpointing,
                        grid_size=64 // downsample_factor, spatial_shapes=spatial_shapes,
                        base_spatial_shape=sd_shapes[-1], layer_idx=len(channels))
        self.mmfs_mid_block = mid
        self._reset_parameters()

    def _reset_parameters(self):
        for blk in self.mmfs_down_blocks:
            blk._reset_parameters()
        self.mmfs_mid_block._reset_parameters()

    def _blocks(self):
        return list(self.mmfs_down_blocks) + [self.mmfs_mid_block]

    def _can_fuse(self):
        blocks = self._blocks()
        n0, v0 = blocks[0].feat_norm, blocks[0].mmfs.value_proj
        return all(isinstance(b.feat_norm, nn.LayerNorm) and b.feat_norm.elementwise_affine
                   and b.feat_norm.bias is not None
                   and b.feat_norm.normalized_shape == n0.normalized_shape and b.feat_norm.eps == n0.eps
                   and b.mmfs.value_proj.weight.shape == v0.weight.shape
                   # (the batched projection evaluates these layers' mathematics without calling them: only unobserved)
                   and type(b.mmfs.value_proj) is nn.Linear and hook_free(b.mmfs.value_proj) and hook_free(b.feat_norm)
                   for b in blocks)

    def project_features(self, mmfs_features):
        """Per level [B, n, C, h_l, w_l] -> ``ProjectedFeatures``: the bank normalised ONCE, then
        every block's value projection with its LayerNorm affine folded into the weights."""
        shapes = [(f.shape[-2], f.shape[-1]) for f in mmfs_features]
        bank = self._pack(mmfs_features)
        norm = self.mmfs_mid_block.feat_norm
        xhat = F.layer_norm(bank, norm.normalized_shape, None, None, norm.eps)
        return ProjectedFeatures(self._project_all(xhat), bank, shapes, [(f, tensor_version(f)) for f in mmfs_features],
                                 self._projection_weights())

    def _project_all(self, xhat):
        """Every block's value projection of the normalised bank as ONE batched GEMM, [tokens, C] x [n_blocks, C, d_inner]
        (the left operand broadcast over the batch): block k's result is contiguous, and with gradients the 13 weight
        gradients are one batched GEMM too instead of 13 products of a [1024, 43 520] by a [43 520, 1024] matrix that fill
        an eighth of the chip each (profiles/r03bb_train_kernels_cfg4.log: 264 us apiece, 14 % of the MFMA peak).
        Same mathematics as ``value_proj_k(feat_norm_k(bank))``: the affine folds into the weights,
        value_proj_k(g_k * xhat + b_k) = (W_k diag g_k) xhat + (W_k b_k + bias_k)."""
        blocks = self._blocks()
        wt = torch.stack([(b.mmfs.value_proj.weight * b.feat_norm.weight).t() for b in blocks])          # [n_blocks, C, d_inner]
        bias = torch.stack([F.linear(b.feat_norm.bias, b.mmfs.value_proj.weight, b.mmfs.value_proj.bias) for b in blocks])
        ys = _ProjectAll.apply(xhat.reshape(-1, xhat.shape[-1]).to(wt.dtype), wt, bias)                 # n_blocks x [tokens, d_inner]
        return [y.view(*xhat.shape[:-1], -1) for y in ys]

    def _projection_weights(self):
        return (cache_epoch(),) + tuple((p.data_ptr(), tensor_version(p)) for b in self._blocks()
                                        for p in (b.feat_norm.weight, b.feat_norm.bias, b.mmfs.value_proj.weight, b.mmfs.value_proj.bias))

    @staticmethod
    def _pack(mmfs_features):
        """Per level [B, n, C, h, w] -> [B, n, sum_l h*w, C] ("b n c h w -> b n (h w) c" + concatenation,
        sd_mmfs.py:241-245); on the device one transposing pass (csrc/mmfs_bank.hip)."""
        f0 = mmfs_features[0]
        B, n = f0.shape[:2]
        if f0.is_cuda:
            bank = gather_bank([f.flatten(0, 1) for f in mmfs_features], torch.arange(B * n, device=f0.device))
            return bank.view(B, n, *bank.shape[1:])
        return torch.cat([f.flatten(3).transpose(2, 3) for f in mmfs_features], dim=2)

    def clear_feature_cache(self):
        """Forget the projected features -- and every fold / table the blocks keep (``levels.invalidate_caches``)."""
        self.__dict__.pop("_projected", None)
        self.clear_caches()

    def forward(self, sample, down_block_res_samples, mmfs_features, mmfs_mask):
        """sample: mid-block input; down_block_res_samples: the UNet's down residuals;
        mmfs_features: per level [B, n, C, h_l, w_l] (or a ``ProjectedFeatures``); mmfs_mask [B, n]
        -> (sample', tuple of residuals')   (sd_mmfs.py:230-272)."""
        assert len(down_block_res_samples) == len(self.mmfs_down_blocks)
        proj = mmfs_features if isinstance(mmfs_features, ProjectedFeatures) else None
        # Under gradient checkpointing the reference recomputes feat_norm + value_proj inside every block's
        # checkpoint and keeps none of them; handing each block a projected bank as a checkpoint INPUT keeps all 13
        # bank-sized tensors alive through the whole step.  That is what ``project_once_in_training`` does when they are a
        # small part of the free memory (1.16 GB at B = 8: 26 projections and 13 weight-gradient GEMMs saved); else training
        # with checkpointing recomputes every block's projection as the reference does (the un-affined normalisation
        # alone is shared: below).
        ckpt = self.training and torch.is_grad_enabled() and any(b.gradient_checkpointing for b in self._blocks())
        if ckpt and proj is None and self._project_once(mmfs_features):
            ckpt = False                                   # (the projections become checkpoint inputs: below)
        if proj is None and self.fused_schedule and self._can_fuse() and not ckpt:
            keep = self.cache_projected_features and not torch.is_grad_enabled() and not self.training
            proj = self.__dict__.get("_projected") if keep else None
            if proj is None or not proj.matches(mmfs_features, self._projection_weights()):
                proj = self.project_features(mmfs_features)
            if keep:
                self.__dict__["_projected"] = proj
            else:
                self.__dict__.pop("_projected", None)
        normed = None
        if proj is not None:
            bank, shapes, values = proj.bank, proj.shapes, proj.values
        else:
            shapes = [(f.shape[-2], f.shape[-1]) for f in mmfs_features]
            bank = self._pack(mmfs_features)
            values = [None] * (len(self.mmfs_down_blocks) + 1)
            if ckpt and self.fused_schedule and self.share_normalised_bank and self._can_fuse():
                # ... but the normalisation WITHOUT the affine is one bank-sized tensor for all 13 blocks: each block
                # folds its own affine into its projection inside its checkpoint (13 + 13 LayerNorm passes over the bank
                # and 13 LayerNorm backward passes become 1 + 1; the projected banks are still recomputed, not kept)
                norm = self.mmfs_mid_block.feat_norm
                normed = F.layer_norm(bank, norm.normalized_shape, None, None, norm.eps)
        # (the images' ranks among the visible ones depend on the mask only: once for the 13 blocks)
        ranks = self.mmfs_mid_block.mmfs._image_relpos(mmfs_mask, 1) if mmfs_mask.dim() == 2 else None
        new_res = tuple(blk(r, bank, mmfs_mask, shapes, value=v, image_ranks=ranks, residual=r, normed=normed)
                        for r, blk, v in zip(down_block_res_samples, self.mmfs_down_blocks, values))
        sample = self.mmfs_mid_block(sample, bank, mmfs_mask, shapes, value=values[-1], image_ranks=ranks, residual=sample,
                                     normed=normed)
        return sample, new_res
