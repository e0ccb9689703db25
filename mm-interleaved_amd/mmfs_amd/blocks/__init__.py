"""The callers either side of the op."""
from .llama_mmfs import LlamaMMFSAttention, LlamaMMFSSchedule, MMFSRMSNorm, ProjectedBank  # noqa: F401
from .sd_mmfs import MMFSBlock, MMFSNet  # noqa: F401
