import sys, time, torch
sys.path[:0] = ["/root/repo", "/root/repo/mm-interleaved_amd"]
import bench, MultiScaleDeformableAttention as MSDA
w = dict(bench.WORKLOADS["cfg5_llm_n4"])
value, shapes, start, loc, attn, grad = bench.make_inputs(w, "cuda", 0, visible="causal")
S = value.shape[1]
MSDA.register_level_tables(shapes, start, S) if hasattr(MSDA, "register_level_tables") else None
for lazy in (False, True, False, True):
    MSDA._event_log = log = []
    for _ in range(30):
        MSDA.ms_deform_attn_backward(value, shapes, start, loc, attn, grad, 1, lazy_zero_attn=lazy)
    torch.cuda.synchronize(); MSDA._event_log = None
    ev = {}
    for n, a, b in log[len(log)//3:]:
        ev.setdefault(n, []).append(a.elapsed_time(b) * 1e3)
    print("lazy" if lazy else "full", {k: round(sum(v)/len(v), 1) for k, v in ev.items()})
