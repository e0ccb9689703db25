# mirrors mm_interleaved/models/utils/ops/functions/__init__.py:9
from .ms_deform_attn_func import MSDeformAttnFunction, ms_deform_attn_core_pytorch  # noqa: F401
from .mmfs_plan_func import MMFSPlanFunction, mmfs_plan_supported  # noqa: F401,E402
from .bank_func import BankGatherFunction, bank_gather_supported  # noqa: F401,E402
from .norm_func import RMSNormFunction, rmsnorm_supported  # noqa: F401,E402
