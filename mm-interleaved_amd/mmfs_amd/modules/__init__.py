# mirrors mm_interleaved/models/utils/ops/modules/__init__.py:10 (MMFS) and
# mm_interleaved/models/encoders/vit_adapter/ops/modules/__init__.py (MSDeformAttn)
from .mmfs import MMFS  # noqa: F401
from .ms_deform_attn import MSDeformAttn  # noqa: F401
