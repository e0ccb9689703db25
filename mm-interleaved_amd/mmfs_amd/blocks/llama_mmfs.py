"""LLM-side feature synchronizer block.

``LlamaMMFSAttention`` mirrors the reference class of the same name
(mm_interleaved/models/decoders/modeling_llama_mmfs.py:311-367; helpers ``deform_inputs`` /
``get_reference_points`` at :283-308): RMSNorm on the token stream and on the image
features, MMFS with every token's reference point at the image centre, output gated by
tanh(gate) (gate initialised to 0).  State-dict keys are the reference's:
``gate``, ``norm1.weight``, ``norm2.weight``, ``attn.<MMFS keys>``.

Used by every ``cross_attention_frequency``-th decoder layer
(modeling_llama_mmfs.py:427-434, 581-583); the stock LLaMA layers around it are out of
scope.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..functions.norm_func import rmsnorm, rmsnorm_supported
from ..graphed import graphed_call
from ..levels import CacheInvalidation, cache_epoch, hook_free, make_level_tables, tensor_version
from ..modules.mmfs import MMFS, FoldedLinear


class MMFSRMSNorm(nn.Module):
    """T5/LLaMA RMS norm: statistics in fp32, cast back to the weight's dtype when that is
    16-bit (modeling_llama_mmfs.py:53-70).  Single parameter ``weight``."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    fused = True          # one gfx950 kernel each way where it applies (csrc/mmfs_norm.hip); off: framework ops
    _behaviour_flags = ("fused", "variance_epsilon")

    def forward(self, x):
        if self.fused and rmsnorm_supported(x, self.weight):
            return rmsnorm(x, self.weight, self.variance_epsilon)
        var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        x = x * torch.rsqrt(var + self.variance_epsilon)
        if self.weight.dtype in (torch.float16, torch.bfloat16):
            x = x.to(self.weight.dtype)
        return self.weight * x


_centre = {}


def centre_reference_points(n_tokens, device):
    """[1, n_tokens, 1, 2] filled with (0.5, 0.5): the pixel centre of a 1x1 grid, what the
    reference's get_reference_points([(1, 1)]) yields, repeated per token (:306-307).  One 2-element tensor per
    device, made once (outside any inference mode: it outlives the call), expanded per call."""
    key = str(torch.device(device))
    base = _centre.get(key)
    if base is None:
        with torch.inference_mode(False), torch.no_grad():
            base = _centre[key] = torch.full((1, 1, 1, 2), 0.5, dtype=torch.float32, device=device)
    return base.expand(1, n_tokens, 1, 2)


def deform_inputs(hidden_states, vision_hidden_states, spatial_shapes=((16, 16),)):
    """(reference_points, spatial_shapes, level_start_index) for a [B, n, hw, C] feature bank:
    the per-image shapes repeated once per image (modeling_llama_mmfs.py:298-308)."""
    _, n, hw, _ = vision_hidden_states.shape
    per_image = sum(int(h) * int(w) for h, w in spatial_shapes)
    repeat = (n * hw) // per_image
    shapes, start, _ = make_level_tables(spatial_shapes, repeat, hidden_states.device)
    return centre_reference_points(hidden_states.size(1), hidden_states.device), shapes, start


class LlamaMMFSAttention(CacheInvalidation, nn.Module):
    _behaviour_flags = ("fold_gate", "spatial_shapes")      # (part of a recorded call's key: mmfs_amd/graphed.py)

    def __init__(self, config, layer_idx):
        super().__init__()
        self.layer_idx = layer_idx
        self.config = config
        self.spatial_shapes = [(s, s) for s in config.spatial_shapes]
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.max_position_embeddings = getattr(config, "max_position_embeddings", None)
        self.vision_hidden_size = config.image_embed_dim
        self.gate = nn.Parameter(torch.tensor([0.0]))
        # hyper-parameters are hard-coded at the call site in the reference (:326-339)
        self.attn = MMFS(
            layer_idx=layer_idx,
            d_model=self.hidden_size,
            d_query=self.hidden_size,
            d_value=self.vision_hidden_size,
            d_out=self.hidden_size,
            n_levels=len(config.spatial_shapes),
            n_heads=16,
            n_points=8,
            ratio=self.vision_hidden_size / self.hidden_size,
            offset_init_magnitude=3.0,
            spatial_shapes=config.spatial_shapes,
            max_num_image_per_seq=50,
        )
        eps = getattr(config, "rms_norm_eps", 1e-6)
        self.norm1 = MMFSRMSNorm(config.hidden_size, eps=eps)
        self.norm2 = MMFSRMSNorm(self.vision_hidden_size, eps=eps)
        self.fold_gate = True                     # no-grad calls: tanh(gate) folded into the output projection (forward)
        self.graph_training_calls = True          # training mode: calls of a repeating shape replay HIP graphs (forward)
        self._gate_fold = FoldedLinear()

    def forward(self, hidden_states, vision_hidden_states=None, cross_attention_mask=None, value=None, image_ranks=None,
                residual=None):
        """hidden_states [B, Lq, hidden]; vision_hidden_states [B, n, sum hw, image_embed_dim];
        cross_attention_mask [B, Lq', n] (float, 1 = visible) -> [B, Lq, hidden].
        ``value`` (an addition): this layer's ``value_proj(norm2(vision_hidden_states))`` [B, n, sum hw, d_inner] as
        a ``LlamaMMFSSchedule`` projected it for all layers at once; the bank is then only looked at for its shape.
        ``image_ranks`` (another): ``LlamaMMFSSchedule.image_ranks(cross_attention_mask, Lq)``, made once per step.
        ``residual`` (a third) [B, Lq, hidden]: the result is ``residual + layer(...)`` -- the decoder layer's own next
        statement (modeling_llama_mmfs.py:700-717), which without gradients rides in the output projection's kernel."""
        args = (hidden_states, vision_hidden_states, cross_attention_mask, value, image_ranks, residual)
        if self.graph_training_calls and self.training:
            # a training step's call as HIP graphs once its shapes have been seen a few times (mmfs_amd/graphed.py) -- the
            # forward with its saved activations, the backward; and the no-grad forward of a checkpointing caller
            return graphed_call(self, self._forward, args, recompute=False, plain=lambda: self._forward(*args))
        return self._forward(*args)

    def _forward(self, hidden_states, vision_hidden_states, cross_attention_mask, value, image_ranks, residual):
        hidden_states = self.norm1(hidden_states)
        if value is None:
            vision_hidden_states = self.norm2(vision_hidden_states)
        ref, shapes, start = deform_inputs(hidden_states, vision_hidden_states, self.spatial_shapes)
        # (not under autocast: there ``out * tanh(gate)`` is a bf16 x fp32 product with an fp32 result, modeling_llama_mmfs.py:356)
        proj = self.attn.output_proj
        if (self.fold_gate and not self.training and not torch.is_grad_enabled() and not torch.is_autocast_enabled()
                and type(proj) is nn.Linear and hook_free(proj)):
            # tanh(gate) * output_proj(x) = ((tanh(gate) W) x + tanh(gate) b): without gradients the gate rides in the
            # output projection's weights (kept until a parameter moves) -- one full-size multiply per layer less
            folded = self._gate_fold.get(proj.weight, proj.bias, self._gate(), None)
            return self.attn(query=hidden_states, reference_points=ref, input_flatten=vision_hidden_states,
                             input_spatial_shapes=shapes, input_level_start_index=start, input_padding_mask=None,
                             attention_mask=cross_attention_mask, value=value, image_ranks=image_ranks,
                             output_weights=folded, output_residual=residual)
        if residual is not None:
            # the gate and the residual sum ride with the output projection (``GatedProjectionFunction`` with gradients)
            return self.attn(query=hidden_states, reference_points=ref, input_flatten=vision_hidden_states,
                             input_spatial_shapes=shapes, input_level_start_index=start, input_padding_mask=None,
                             attention_mask=cross_attention_mask, value=value, image_ranks=image_ranks,
                             output_gate=self._gate(), output_residual=residual)
        out = self.attn(query=hidden_states, reference_points=ref, input_flatten=vision_hidden_states,
                        input_spatial_shapes=shapes, input_level_start_index=start,
                        input_padding_mask=None, attention_mask=cross_attention_mask, value=value, image_ranks=image_ranks)
        return out * self._gate()

    def _gate(self):
        """tanh(gate) (modeling_llama_mmfs.py:356): a one-element kernel per layer and step; without gradients it is
        kept until the parameter moves."""
        if torch.is_grad_enabled() or self.training:
            return self.gate.tanh()
        sig = (cache_epoch(), self.gate.data_ptr(), tensor_version(self.gate), self.gate.dtype, torch.is_inference_mode_enabled())
        hit = getattr(self, "_gate_tanh", None)
        if hit is None or hit[0] != sig:
            hit = self._gate_tanh = (sig, self.gate.tanh())
        return hit[1]


class ProjectedBank:
    """``value_proj_k(norm2_k(bank))`` of every MMFS layer of a decoder for one feature bank: ``values[k]`` is
    layer k's [B, n, sum hw, d_inner] (contiguous).  Made by ``LlamaMMFSSchedule.project``."""

    def __init__(self, values, bank, source=None, weights=None):
        self.values, self.bank = values, bank
        self.source, self.weights = source, weights      # identity cache: (tensor, version), parameter signature

    def matches(self, bank, weights):
        return (self.source is not None and bank is self.source[0] and tensor_version(bank) == self.source[1]
                and weights == self.weights)


class LlamaMMFSSchedule:
    """The MMFS layers of a decoder, organised once (SURVEY 8f N3's idea on the LLM side).

    Every ``cross_attention_frequency``-th decoder layer owns a ``LlamaMMFSAttention`` (8 of 32 at 7B, 10 of 40
    at 13B: modeling_llama_mmfs.py:581-583) and ALL of them RMS-normalise and ``value_proj`` the SAME feature bank
    (:352-353; ops/modules/mmfs.py:165); while generating, the bank does not even change between decode steps
    (mm_interleaved.py:598-664, utils/causal_lm_cascade.py:139-153).  Same mathematics:

      * one normalisation: RMS statistics do not depend on the layer, only the gain does, and it folds into the
        projection: ``value_proj_k(g_k * xhat) = (W_k diag g_k) xhat + b_k``;
      * one batched GEMM ``[tokens, d_value] x [n_layers, d_value, d_inner]`` for all layers' projections (each
        layer's result contiguous);
      * outside autograd the projections are kept for as long as the caller passes the same, unmodified bank
        tensor and the parameters that went in do not move (identity + version counters): a decode step then
        runs no bank-sized kernel at all.

    ``schedule = LlamaMMFSSchedule(layers)``; per forward of the decoder ``bank = schedule.project(features)``
    and layer k is called as ``layers[k](hidden, features, mask, value=bank.values[k])``.
    Differentiable: gradients reach every ``W_k``, ``b_k``, ``g_k`` and the features."""

    cache_projected_bank = True

    def __init__(self, layers):
        self.layers = list(layers)
        assert self.layers and all(isinstance(l, LlamaMMFSAttention) for l in self.layers)
        self._projected = None

    def image_ranks(self, cross_attention_mask, n_queries):
        """The images' ranks among the visible ones, per token (ops/modules/mmfs.py:154-163): a function of the mask
        only, the same for every layer of a step -- half a dozen small integer kernels once instead of once per layer.
        Pass it to each layer as ``image_ranks=``."""
        return self.layers[0].attn._image_relpos(cross_attention_mask, n_queries)

    def can_fuse(self):
        n0, v0 = self.layers[0].norm2, self.layers[0].attn.value_proj
        return all(l.norm2.weight.shape == n0.weight.shape and l.norm2.variance_epsilon == n0.variance_epsilon
                   and l.attn.value_proj.weight.shape == v0.weight.shape
                   and (l.attn.value_proj.bias is None) == (v0.bias is None)
                   # (the batched projection evaluates these layers' mathematics without calling them: only unobserved)
                   and type(l.attn.value_proj) is nn.Linear and hook_free(l.attn.value_proj) and hook_free(l.norm2)
                   for l in self.layers)

    def _weights(self):
        return (cache_epoch(),) + tuple((p.data_ptr(), tensor_version(p)) for l in self.layers
                                        for p in (l.norm2.weight, l.attn.value_proj.weight, l.attn.value_proj.bias) if p is not None)

    def _project(self, bank):
        norm = self.layers[0].norm2
        var = bank.to(torch.float32).pow(2).mean(-1, keepdim=True)
        xhat = bank * torch.rsqrt(var + norm.variance_epsilon)
        if norm.weight.dtype in (torch.float16, torch.bfloat16):
            xhat = xhat.to(norm.weight.dtype)
        # [n_layers, d_value, d_inner]: W_k^T with the gain folded into its rows
        wt = torch.stack([(l.attn.value_proj.weight * l.norm2.weight).t() for l in self.layers])
        y = torch.matmul(xhat.reshape(1, -1, xhat.shape[-1]).to(wt.dtype), wt)         # [n_layers, tokens, d_inner]
        if self.layers[0].attn.value_proj.bias is not None:
            y = y + torch.stack([l.attn.value_proj.bias for l in self.layers])[:, None, :]
        return [y[k].view(*bank.shape[:-1], -1) for k in range(len(self.layers))]

    def project(self, vision_hidden_states):
        """[B, n, sum hw, image_embed_dim] -> ``ProjectedBank`` (kept across calls outside autograd)."""
        if not self.can_fuse():         # the reference's schedule, layer by layer
            return ProjectedBank([l.attn.value_proj(l.norm2(vision_hidden_states)) for l in self.layers],
                                 vision_hidden_states)
        keep = self.cache_projected_bank and not torch.is_grad_enabled() and not any(l.training for l in self.layers)
        sig = self._weights()
        if keep and self._projected is not None and self._projected.matches(vision_hidden_states, sig):
            return self._projected
        proj = ProjectedBank(self._project(vision_hidden_states), vision_hidden_states,
                             (vision_hidden_states, tensor_version(vision_hidden_states)), sig)
        self._projected = proj if keep else None
        return proj

    def clear_cache(self):
        """Forget the projected bank -- and every fold / table the layers keep (``levels.invalidate_caches``)."""
        self._projected = None
        for l in self.layers:
            l.clear_caches()
