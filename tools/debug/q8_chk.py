import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import MultiScaleDeformableAttention as MSDA
from helpers import make_inputs
buf = (ctypes.c_uint * 16)()
MSDA._lib.mmfs_debug_q8_check(buf, 1)
dev = lambda t, dt: t.to("cuda", dt) if t.is_floating_point() else t.to("cuda")
for rep in range(6):
    x = make_inputs(1, 8, 32, 64, 8, [(32, 32), (16, 16), (8, 8)], seed=21 + rep, loc_range=(0.05, 0.95), dtype=torch.float16)
    MSDA._fwd_algo = "gather"
    g = MSDA.ms_deform_attn_forward(*[dev(x[k], torch.float16) for k in ("value", "shapes", "start", "loc", "attn")], 1).double().cpu().numpy()
    MSDA._fwd_algo = "slices"
    a = MSDA.ms_deform_attn_forward(*[dev(x[k], torch.float16) for k in ("value", "shapes", "start", "loc", "attn")], 1).double().cpu().numpy()
    torch.cuda.synchronize()
    MSDA._lib.mmfs_debug_q8_check(buf, 1)
    print("rep %d: max err %.3e; mismatches: prefetched loc %d, attn %d | own offset read back %d | B offsets q even %d, q odd %d, (of them kb 3: %d)"
          % (rep, np.abs(a - g).max(), buf[0], buf[1], buf[2], buf[3], buf[4], buf[5]))
