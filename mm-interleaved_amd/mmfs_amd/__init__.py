"""mmfs_amd -- host-side mirror of the reference's MMFS operator interface, backed
by the hand-written gfx950 kernels of libmmfs_msda.so.

Layout mirrors mm_interleaved/models/utils/ops/ of the reference:
    mmfs_amd.functions   MSDeformAttnFunction, ms_deform_attn_core_pytorch
    mmfs_amd.modules     MMFS, MSDeformAttn
and the callers either side of the op:
    mmfs_amd.blocks      LlamaMMFSAttention, MMFSBlock, MMFSNet
    mmfs_amd.bank        feature-bank builders + the RCCL all-gather of image features
"""
from .functions import MSDeformAttnFunction, ms_deform_attn_core_pytorch  # noqa: F401
from .levels import invalidate_caches  # noqa: F401

__version__ = "0.1.0"
