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
from torch import nn

from ..levels import make_level_tables
from ..modules.mmfs import MMFS


class MMFSRMSNorm(nn.Module):
    """T5/LLaMA RMS norm: statistics in fp32, cast back to the weight's dtype when that is
    16-bit (modeling_llama_mmfs.py:53-70).  Single parameter ``weight``."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        x = x * torch.rsqrt(var + self.variance_epsilon)
        if self.weight.dtype in (torch.float16, torch.bfloat16):
            x = x.to(self.weight.dtype)
        return self.weight * x


def centre_reference_points(n_tokens, device):
    """[1, n_tokens, 1, 2] filled with (0.5, 0.5): the pixel centre of a 1x1 grid, what the
    reference's get_reference_points([(1, 1)]) yields, repeated per token (:306-307)."""
    return torch.full((1, 1, 1, 2), 0.5, dtype=torch.float32, device=device).expand(1, n_tokens, 1, 2)


def deform_inputs(hidden_states, vision_hidden_states, spatial_shapes=((16, 16),)):
    """(reference_points, spatial_shapes, level_start_index) for a [B, n, hw, C] feature bank:
    the per-image shapes repeated once per image (modeling_llama_mmfs.py:298-308)."""
    _, n, hw, _ = vision_hidden_states.shape
    per_image = sum(int(h) * int(w) for h, w in spatial_shapes)
    repeat = (n * hw) // per_image
    shapes, start, _ = make_level_tables(spatial_shapes, repeat, hidden_states.device)
    return centre_reference_points(hidden_states.size(1), hidden_states.device), shapes, start


class LlamaMMFSAttention(nn.Module):
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

    def forward(self, hidden_states, vision_hidden_states=None, cross_attention_mask=None):
        """hidden_states [B, Lq, hidden]; vision_hidden_states [B, n, sum hw, image_embed_dim];
        cross_attention_mask [B, Lq', n] (float, 1 = visible) -> [B, Lq, hidden]."""
        hidden_states = self.norm1(hidden_states)
        vision_hidden_states = self.norm2(vision_hidden_states)
        ref, shapes, start = deform_inputs(hidden_states, vision_hidden_states, self.spatial_shapes)
        out = self.attn(query=hidden_states, reference_points=ref, input_flatten=vision_hidden_states,
                        input_spatial_shapes=shapes, input_level_start_index=start,
                        input_padding_mask=None, attention_mask=cross_attention_mask)
        return out * self.gate.tanh()
