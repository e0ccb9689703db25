"""MMFS -- Multi-image Multi-scale Feature Synchronizer, host side.

Same constructor, parameters (names and shapes -> checkpoint compatible),
``forward`` signature and results as the reference module
(mm_interleaved/models/utils/ops/modules/mmfs.py:26-276).  The sampling itself runs in
the gfx950 kernels (``MSDeformAttnFunction``); the dense projections go to hipBLASLt via
``F.linear`` (MFMA) -- they are plain contractions.

What is organised differently from the reference forward, with identical mathematics:

* the reference repeats the query ``n_images`` times and pushes the copies through the
  d_query x d_query ``dynamic_offset_mask`` GEMM (mmfs.py:174-175); the only per-image
  term is the additive ``query_relpos`` row (mmfs.py:177-179), and the two heads that
  follow are linear, so
      head(W q + r_k) = head(W q) + (head.weight @ r_k)
  -> one GEMM on the un-repeated query plus a [max_images, out] table lookup;
* no device->host synchronisation: the reference's three asserts on device tensors and
  its ``torch.nonzero`` (mmfs.py:147-149, 177, 209-211 / 220-222) each stall the stream;
  the nonzero only feeds a write that line 225 overwrites;
* the sink ("ignore") slot is handled as the constant logit it is (mmfs.py:225).
"""
import math
import os

import torch
import torch.nn.functional as F
from torch import nn

from ..functions import MSDeformAttnFunction
from ..functions.linear_func import small_linear, token_linear
from ..functions.block_func import GatedProjectionFunction, IgnoreTokenFunction
from ..functions.mmfs_plan_func import MMFSHeadsPlanFunction, MMFSPlanFunction, mmfs_plan_supported, mmfs_sample_forward
from ..levels import CacheInvalidation, cache_epoch, hook_free, host_shapes, tensor_version


class FoldedLinear:
    """``outer(inner(x))`` of two Linear layers with nothing non-linear between them as ONE Linear layer, for calls that
    want no gradients: W = W_outer W_inner, b = W_outer b_inner + b_outer (``outer`` may also be a scalar tensor: a gate).
    Formed once in fp32 (fp64 for fp64 parameters), rounded once to the storage type, kept until one of the parameters
    moves (data pointer + version counter, as ``MMFS._plan_tables``) or the package's cache epoch does (mode changes,
    state-dict loads, ``clear_caches()``: ``levels.invalidate_caches`` -- writes through ``.data`` move no counter); made
    in inference mode it is not used outside it.
    Same mathematics as the two layers, other rounding points: the intermediate is not rounded, the product is."""

    def __init__(self):
        self._kept = None

    def get(self, inner_w, inner_b, outer_w, outer_b):
        ps = tuple(t for t in (inner_w, inner_b, outer_w, outer_b) if t is not None)
        # (an operand made inside inference mode -- a kept tanh(gate) -- has no version counter: it cannot change either)
        sig = (cache_epoch(), torch.is_inference_mode_enabled()) + tuple(
            (t.data_ptr(), tensor_version(t), t.dtype) for t in ps)
        if self._kept is not None and self._kept[0] == sig:
            return self._kept[1]
        dt = inner_w.dtype
        ft = torch.promote_types(dt, torch.float32)
        wi = inner_w.to(ft)
        bi = inner_b.to(ft) if inner_b is not None else None
        if outer_w.dim() <= 1 and outer_w.numel() == 1:                  # a scalar in front: gate * (W x + b)
            g = outer_w.to(ft).reshape(())
            w, b = g * wi, (g * bi if bi is not None else None)
        else:
            wo = outer_w.to(ft)
            w = wo @ wi
            b = wo @ bi if bi is not None else None
            if outer_b is not None:
                b = outer_b.to(ft) if b is None else b + outer_b.to(ft)
        res = (w.to(dt), b.to(dt) if b is not None else None)
        self._kept = (sig, res)
        return res


class MMFS(CacheInvalidation, nn.Module):
    # the switches that choose this module's path: part of the key of a recorded call (mmfs_amd/graphed.py)
    _behaviour_flags = ("stack_heads_in_training", "fused_plan", "fold_query_projection", "fused_sampler", "im2col_step",
                        "max_num_image_per_seq")

    def __init__(
        self,
        layer_idx=0,
        d_model=256,
        d_query=-1,
        d_value=256,
        d_out=-1,
        n_levels=4,
        n_heads=8,
        n_points=8,
        ratio=1.0,
        offset_init_magnitude=3,
        spatial_shapes=[16],
        base_spatial_shape=16,
        max_num_image_per_seq=50,
    ):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError(f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        d_query = d_model if d_query < 0 else d_query
        d_out = d_model if d_out < 0 else d_out
        if len(spatial_shapes) != n_levels:
            raise AssertionError("one spatial shape per level")

        self.layer_idx = layer_idx
        self.im2col_step = 1                      # kept for API fidelity; the HIP op ignores it
        self.d_model = d_model
        self.n_levels = n_levels
        self.n_heads = n_heads
        self.n_points = n_points
        # with gradients: the two query heads as one GEMM on stacked weights, the two tables as another (_plan_tables)
        self.stack_heads_in_training = True
        self.ratio = ratio
        self.offset_init_magnitude = offset_init_magnitude
        self.max_num_image_per_seq = max_num_image_per_seq
        self.fused_plan = True                    # use csrc/mmfs_plan.hip when it applies
        self.fold_query_projection = True         # no-grad calls: dynamic_offset_mask folded into the heads (_plan_tables)
        # inference: plan -> sampler in one kernel, no loc / attn tensors (MMFS_FUSED_SAMPLER=0: measurements)
        self.fused_sampler = os.environ.get("MMFS_FUSED_SAMPLER", "1") != "0"
        d_inner = int(d_model * ratio)
        self.d_inner = d_inner

        self.register_buffer("scale_ratios",
                             torch.tensor([s / base_spatial_shape for s in spatial_shapes]),
                             persistent=False)
        # parameter names/shapes as mmfs.py:85-96
        self.sampling_offsets = nn.Linear(d_query, n_heads * n_points * 2)
        self.ignore_token = nn.Parameter(torch.zeros(1, 1, 1, d_inner), requires_grad=False)
        self.dynamic_offset_mask = nn.Linear(d_query, d_query)
        self.attention_weights = nn.Linear(d_query, n_heads * n_levels * (n_points + 1))
        self.value_proj = nn.Linear(d_value, d_inner)
        self.output_proj = nn.Linear(d_inner, d_out)
        self.query_relpos = nn.Embedding(max_num_image_per_seq, d_query)
        self._tables = None                       # (parameter signature, what _plan_tables made of them) -- no-grad calls only
        self._reset_parameters()

    def _reset_parameters(self):
        """Initialisation of mmfs.py:102-118 (called again by the model builders)."""
        m = self.offset_init_magnitude
        with torch.no_grad():
            self.sampling_offsets.weight.zero_()
            self.sampling_offsets.bias.copy_(
                torch.empty(self.n_heads * self.n_points * 2).uniform_(-m, m))
            self.attention_weights.bias.zero_()
            nn.init.xavier_uniform_(self.value_proj.weight)
            self.value_proj.bias.zero_()
            nn.init.xavier_uniform_(self.output_proj.weight)
            self.output_proj.bias.zero_()
            self.dynamic_offset_mask.bias.zero_()
            nn.init.trunc_normal_(self.query_relpos.weight, std=0.02)

    # ------------------------------------------------------------------ pieces
    def _image_relpos(self, attention_mask, Lq):
        """Newest visible image -> 1, older -> 2, 3, ...; invisible -> 0 (mmfs.py:154-163).
        Returns [N, 1 or Lq, n] (long).  It depends on the mask only: the MMFS layers of a decoder, which are all
        handed the same mask in a step (modeling_llama_mmfs.py:352-353), can share it -- ``forward(...,
        image_ranks=...)``, ``LlamaMMFSSchedule.image_ranks``."""
        m = attention_mask.long()
        relpos = (m.sum(-1, keepdim=True) + 1 - m.cumsum(-1)) * m
        if relpos.dim() == 2:
            return relpos[:, None, :]
        if relpos.shape[1] != Lq:                 # decode step: mask still has the whole history
            relpos = relpos[:, -1:, :]
        return relpos

    def _plan_tables(self, fused):
        """What the plan needs that depends on the PARAMETERS only: the relative-position table pushed through the two
        heads (``off_tab`` [max_img, H*P*2]; ``att_tab`` -- all H*L*(P+1) columns for the framework-op statement, the
        P point columns for the fused kernel) and, for the fused kernel, the point columns of the attention head
        itself.  With gradients they are part of the graph and made per call (mmfs.py:174-176 evaluates them inside
        head(q + table[relpos])); without (sampling, decoding) they are kept until a parameter moves -- three of a
        layer's six small GEMMs and two weight-slicing copies per decode step."""
        # (kept in eval mode only: a training run updates parameters in ways no counter sees -- levels.invalidate_caches)
        keep = not torch.is_grad_enabled() and not self.training
        sig = None
        if keep:
            ps = (self.query_relpos.weight, self.sampling_offsets.weight, self.sampling_offsets.bias,
                  self.attention_weights.weight, self.attention_weights.bias,
                  self.dynamic_offset_mask.weight, self.dynamic_offset_mask.bias)
            # (whether a hook could observe ``dynamic_offset_mask`` is part of the key: tables made while a profiler's
            # or a FLOP counter's process-wide hooks were registered keep the two GEMMs, and must not outlive the hooks
            # -- nor may folded ones be used while hooks are there: tools/module_bench.py measured both, r04zf)
            sig = (cache_epoch(), self.fold_query_projection and hook_free(self.dynamic_offset_mask), torch.is_autocast_enabled())
            sig = sig + (fused, torch.is_inference_mode_enabled()) + tuple((t.data_ptr(), tensor_version(t), t.dtype) for t in ps if t is not None)
            if self._tables is not None and self._tables[0] == sig:
                return self._tables[1]
        table = self.query_relpos.weight                                      # [max_img, d_query]
        off_tab = F.linear(table, self.sampling_offsets.weight) if (keep or not fused or not self.stack_heads_in_training) else None   # [max_img, H*P*2]
        if fused:
            H, L, P = self.n_heads, self.n_levels, self.n_points
            dq = self.attention_weights.in_features
            aw_w = self.attention_weights.weight.view(H, L, P + 1, dq)[:, :, :P].reshape(H * L * P, dq)
            aw_b = self.attention_weights.bias.view(H, L, P + 1)[:, :, :P].reshape(H * L * P)
            # (kept tables only: the two heads read the same activations -- stacked, they are ONE GEMM per call whose
            # result's two column ranges the fused sampler takes as they lie, ``mmfs_sample_forward_heads``)
            so = self.sampling_offsets
            # (with gradients too: one GEMM forward, two backward instead of two and four, and the plan's Function hands
            # back ONE gradient for the stacked result -- ``stack_heads_in_training``)
            stack = ((keep or self.stack_heads_in_training) and so.bias is not None and so.weight.dtype == aw_w.dtype
                     and so.bias.dtype == aw_b.dtype)
            cat_w = torch.cat((so.weight, aw_w), 0) if stack else None
            cat_b = torch.cat((so.bias, aw_b), 0) if stack else None
            # ... and nothing non-linear stands between ``dynamic_offset_mask`` (d_query x d_query: 32 MB of weights at the
            # LLM's width, two thirds of the bytes a decode step of a layer streams) and the heads:
            #     heads(W_a x + b_a) = (W_h W_a) x + (W_h b_a + b_h)
            # so the kept weights are the PRODUCT (formed once in fp32, rounded once to the storage type) and a call
            # evaluates one [H*P*2 + H*L*P, d_query] GEMM on the query itself.  Same mathematics, other rounding points
            # (the intermediate is not rounded to 16 bits, the folded weights are): ``fold_query_projection = False``
            # keeps the two GEMMs.
            dom = self.dynamic_offset_mask
            fold = (stack and keep and self.fold_query_projection and dom.weight.dtype == cat_w.dtype and not torch.is_autocast_enabled()
                    and type(dom) is nn.Linear and hook_free(dom))
            fold_w = fold_b = None
            if fold:
                ft = torch.promote_types(cat_w.dtype, torch.float32)           # (fp32 for 16-bit storage)
                wh = cat_w.to(ft)
                fold_w = (wh @ dom.weight.to(ft)).to(cat_w.dtype)
                fold_b = cat_b.to(ft) if dom.bias is None else torch.addmv(cat_b.to(ft), wh, dom.bias.to(ft))
                fold_b = fold_b.to(cat_b.dtype)
            if stack and not keep:
                # both tables as one GEMM on the stacked weights; the plan's Function reads the two column ranges
                res = (F.linear(table, cat_w), None, aw_w, aw_b, cat_w, cat_b, None, None)
            else:
                # (the offsets' table was left out above on the assumption that the heads stack; they do not when the two
                # Linear layers differ in dtype or the offsets have no bias: ADVICE r4)
                if off_tab is None:
                    off_tab = F.linear(table, self.sampling_offsets.weight)
                res = (off_tab, F.linear(table, aw_w), aw_w, aw_b, cat_w, cat_b, fold_w, fold_b)
        else:
            res = (off_tab, F.linear(table, self.attention_weights.weight), None, None, None, None, None, None)   # [max_img, H*L*(P+1)]
        if keep:
            self._tables = (sig, res)
        return res

    def sampling_plan(self, query, reference_points, input_spatial_shapes, attention_mask, n_images, sampler=None,
                      image_ranks=None):
        """Everything between the query and the op: sampling locations [N,Lq,H,n*L,P,2],
        attention weights over the real points [N,Lq,H,n*L,P], and the summed sink weights
        [N,Lq,H] (mmfs.py:154-163, 174-265)."""
        N, Lq, _ = query.shape
        H, L, P, n = self.n_heads, self.n_levels, self.n_points, n_images
        nL = n * L
        assert attention_mask.dim() in (2, 3) and attention_mask.shape[-1] == n

        relpos = self._image_relpos(attention_mask, Lq) if image_ranks is None else image_ranks     # [N, 1|Lq, n]
        assert relpos.dim() == 3 and relpos.shape[0] == N and relpos.shape[1] in (1, Lq) and relpos.shape[2] == n
        if n >= self.max_num_image_per_seq:       # only then can an index leave the table
            assert int(relpos.max()) < self.max_num_image_per_seq

        if self.fused_plan and mmfs_plan_supported(query, reference_points, L, P, n):
            # one gfx950 kernel for the rest (csrc/mmfs_plan.hip), fp32 inside, rounded once.  Only
            # the P point columns of the attention head are evaluated: its (P+1)-th column is
            # overwritten by a constant in the reference (mmfs.py:225) and never gets a gradient.
            off_tab, att_tab, aw_w, aw_b, cat_w, cat_b, fold_w, fold_b = self._plan_tables(True)
            both = None
            if fold_w is not None and query.dtype == fold_w.dtype:            # (no gradients wanted: see _plan_tables)
                both = small_linear(query, fold_w, fold_b)                    # [N, Lq, H*P*2 + H*L*P], straight from the query
                off_q, att_q = both[..., :H * P * 2], both[..., H * P * 2:]
            else:
                q = token_linear(query, self.dynamic_offset_mask)             # one GEMM, not n
                if cat_w is not None and q.dtype == cat_w.dtype:
                    both = token_linear(q, cat_w, cat_b)                      # [N, Lq, H*P*2 + H*L*P]
                    off_q, att_q = both[..., :H * P * 2], both[..., H * P * 2:]
                else:
                    off_q, att_q = self.sampling_offsets(q), F.linear(q, aw_w, aw_b)
            if att_tab is None:               # (with gradients: both tables as the column ranges of one GEMM's result)
                tabs = off_tab
                # (the kernels read the stacked rows in 16-byte vectors: both column ranges have to start and stride on
                # 16 bytes -- P = 4 in 16-bit storage with H * (2 + L) odd does not; then the sliced, packed path: ADVICE r4)
                es = both.element_size() if both is not None else 1
                vec_ok = (both is not None and (both.shape[-1] * es) % 16 == 0 and (H * P * 2 * es) % 16 == 0
                          and both.data_ptr() % 16 == 0 and tabs.data_ptr() % 16 == 0)
                if vec_ok and sampler is None and tabs.dtype == both.dtype:
                    return MMFSHeadsPlanFunction.apply(both, tabs, relpos, reference_points[:, :, 0, :],
                                                       input_spatial_shapes, self.scale_ratios, H, L, P)
                off_tab, att_tab = tabs[:, :H * P * 2], tabs[:, H * P * 2:]
            heads = (off_q, att_q, off_tab, att_tab, relpos,
                     reference_points[:, :, 0, :], input_spatial_shapes, self.scale_ratios, H, L, P)
            if sampler is not None:
                # inference: the plan feeds the sampler inside one kernel (``sampler`` = (value, level starts));
                # returns the op's output and the sink weights instead of loc / attn
                # (``sampler[2]``: the ignore token, whose term the kernel then adds itself -- the sink weights come back None)
                res = mmfs_sample_forward(sampler[0], input_spatial_shapes, sampler[1], *heads[:4], relpos,
                                          heads[5], self._ratios32(), H, L, P, token=sampler[2])
                if res is not None:
                    return None, res[0], (res[1] if sampler[2] is None else None)
            loc, attn, sink_sum = MMFSPlanFunction.apply(off_q.contiguous(), att_q.contiguous(), *heads[2:])
            return loc, attn, sink_sum

        q = self.dynamic_offset_mask(query)                                   # one GEMM, not n
        off_tab, att_tab = self._plan_tables(False)[:2]
        # offsets: [N, Lq, 1, :] + [N, 1|Lq, n, :]  ->  [N, Lq, n, H, P, 2]
        offsets = self.sampling_offsets(q)[:, :, None, :] + off_tab[relpos]
        offsets = offsets.view(N, Lq, n, H, 1, P, 2) * self.scale_ratios.view(1, 1, 1, 1, L, 1, 1).to(offsets.dtype)
        offsets = offsets.permute(0, 1, 3, 2, 4, 5, 6).reshape(N, Lq, H, nL, P, 2)

        logits = self.attention_weights(q)[:, :, None, :] + att_tab[relpos]   # [N, Lq, n, H*L*(P+1)]
        logits = logits.view(N, Lq, n, H, L, P + 1).permute(0, 1, 3, 2, 4, 5)  # [N, Lq, H, n, L, P+1]
        # invisible images get -10000 on every point logit (mmfs.py:203-207, 213-218) ...
        am = attention_mask
        if am.dim() == 3 and am.shape[1] != Lq:
            am = am[:, -1:, :]
        penalty = (1.0 - am.to(logits.dtype)) * -10000.0
        penalty = penalty[:, None, None, :] if am.dim() == 2 else penalty[:, :, None, :]
        points = logits[..., :P] + penalty[..., None, None]
        # ... and every level's sink slot is the constant -log(n*L) (mmfs.py:225)
        sink = points.new_full((N, Lq, H, n, L, 1), -math.log(nL))
        probs = F.softmax(torch.cat((points, sink), -1).reshape(N, Lq, H, nL * (P + 1)), -1)
        probs = probs.view(N, Lq, H, nL, P + 1)
        attn, sink_w = probs[..., :P].contiguous(), probs[..., P].sum(-1)

        if reference_points.shape[-1] == 2:
            wh = torch.stack((input_spatial_shapes[:, 1], input_spatial_shapes[:, 0]), -1)
            loc = reference_points[:, :, None, :, None, :] + offsets / wh[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            loc = reference_points[:, :, None, :, None, :2] + \
                offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError(
                f"Last dim of reference_points must be 2 or 4, but get {reference_points.shape[-1]} instead.")
        return loc, attn, sink_w

    def _ratios32(self):
        """``scale_ratios`` in fp32, as the fused sampler reads it: a module in 16-bit storage holds the buffer in 16 bits,
        and a decode step converted it once per layer (a kernel of its own, 8 per step).  Kept like the tables."""
        r = self.scale_ratios
        if r.dtype == torch.float32:
            return r
        if torch.is_grad_enabled() or self.training:
            return r.float()
        sig = (cache_epoch(), r.data_ptr(), tensor_version(r), torch.is_inference_mode_enabled())
        hit = self.__dict__.get("_ratios_f32")
        if hit is None or hit[0] != sig:
            hit = self.__dict__["_ratios_f32"] = (sig, r.float())
        return hit[1]

    def _project_out(self, out, output_weights, residual=None, gate=None):
        """``output_proj`` (or the caller's folded weights in its place), ``+ residual`` if the caller handed one; a
        handful of token rows without gradients: the weight-streaming kernel (functions/linear_func.py), the residual
        added in its store."""
        if gate is not None:                      # (a caller's one-element gate on the projected output: with gradients)
            proj = self.output_proj
            if (residual is not None and torch.is_grad_enabled() and not torch.is_autocast_enabled() and type(proj) is nn.Linear
                    and hook_free(proj) and gate.dtype == out.dtype == residual.dtype == proj.weight.dtype):
                return GatedProjectionFunction.apply(out, proj.weight, proj.bias, gate, residual)
            y = proj(out) * gate
            return y if residual is None else residual + y
        if output_weights is None:
            proj = self.output_proj
            if torch.is_grad_enabled() or type(proj) is not nn.Linear or not hook_free(proj):
                y = token_linear(out, proj)       # (with gradients: a layer that is wrapped / hooked is called as a layer)
                return y if residual is None else residual + y
            output_weights = (proj.weight, proj.bias)
        return small_linear(out, *output_weights, residual=residual)

    # ------------------------------------------------------------------ forward
    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None, attention_mask=None, value=None, image_ranks=None,
                output_weights=None, output_residual=None, output_gate=None):
        """Arguments and result as mmfs.py:120-141 (``value`` is an addition: the caller's own
        ``value_proj(input_flatten)`` [N, n, hw, d_inner], e.g. one an ``MMFSNet`` projected for
        all its blocks at once; ``input_flatten`` is then only looked at for its shape; ``image_ranks`` another: this
        module's ``_image_relpos(attention_mask, Lq)`` as a caller made it once for several layers; ``output_weights``
        a third: (weight, bias) to use in ``output_proj``'s place -- a caller's ``FoldedLinear`` of it with what follows;
        ``output_residual`` [N, Lq, d_out] a fourth: added to the projected output -- the decoder layer's
        ``residual + hidden_states``, which a decode step then gets inside the projection's kernel; ``output_gate`` [1] a
        fifth: multiplies the projected output before that sum -- the LLM layer's tanh(gate), which with gradients becomes
        part of the projection's node, ``GatedProjectionFunction``):
        query [N, Lq, d_query]; reference_points [N|1, Lq, 1|n*L, 2|4] in [0,1];
        input_flatten [N, n_images, sum_l H_l*W_l, d_value]; input_spatial_shapes [n*L, 2];
        input_level_start_index [n*L]; input_padding_mask [N, n, hw] or None;
        attention_mask [N, n] or [N, Lq, n]  ->  [N, Lq, d_out]."""
        N, Lq, _ = query.shape
        N, n, hw, _ = input_flatten.shape
        host = host_shapes(input_spatial_shapes)
        if host is not None:                      # the reference checks this on the device (sync)
            assert int((host[:, 0] * host[:, 1]).sum()) == n * hw, (host.tolist(), n * hw)
        assert input_spatial_shapes.shape[0] == n * self.n_levels

        if value is None:
            value = self.value_proj(input_flatten)
        else:
            assert value.shape == (N, n, hw, self.d_inner), (value.shape, (N, n, hw, self.d_inner))
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
        value = value.reshape(N, n * hw, self.n_heads, self.d_inner // self.n_heads).contiguous()

        # no autograd graph wanted (sampling / decoding): plan and sampler run as one kernel, the locations
        # and weights never exist as tensors (csrc/mmfs_plan.hip, mmfs_sample_fwd); bit-identical output
        fuse = (self.fused_sampler and self.fused_plan and value.is_cuda and query.dtype == value.dtype
                and not (torch.is_grad_enabled() and (query.requires_grad or value.requires_grad
                                                      or any(p.requires_grad for p in self.parameters()))))
        # (the ignore token's term inside the fused kernel where the types agree: else three framework kernels below)
        tok_in = self.ignore_token if (fuse and self.ignore_token.dtype == value.dtype) else None
        loc, attn, sink_w = self.sampling_plan(query, reference_points, input_spatial_shapes, attention_mask, n,
                                               sampler=(value, input_level_start_index, tok_in) if fuse else None,
                                               image_ranks=image_ranks)
        if loc is None:
            out = attn                            # (the fused kernel's result)
            if sink_w is None:                    # ... the ignore token's term included
                return self._project_out(out, output_weights, output_residual, output_gate)
        else:
            # (last argument: the softmax that made ``attn`` multiplies the gradient of every weight by the
            # weight itself, so the op need not compute it where the weight -- an invisible image -- is 0)
            out = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                             loc.to(value.dtype).contiguous(), attn, self.im2col_step, True)
        # the sinks' share goes to the (frozen, zero-initialised) ignore token (mmfs.py:236-241, 274)
        tok = self.ignore_token.view(1, 1, self.n_heads, -1)
        if (out.is_cuda and torch.is_grad_enabled() and not torch.is_autocast_enabled() and tok.dtype == out.dtype
                and out.dtype in (torch.float16, torch.bfloat16) and out.is_contiguous()):
            # one product each way instead of four passes over [tokens, d_inner] tensors (functions/block_func.py)
            out = IgnoreTokenFunction.apply(out.view(N * Lq, -1), tok.view(self.n_heads, -1),
                                            sink_w.reshape(N * Lq, self.n_heads)).view(N, Lq, -1)
        else:
            out = out + (tok * sink_w[..., None].to(tok.dtype)).reshape(N, Lq, -1).to(out.dtype)
        return self._project_out(out, output_weights, output_residual, output_gate)
