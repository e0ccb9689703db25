#!/usr/bin/env python3
"""Small steps replayed from a HIP graph (host floor): is a forward + backward of a SMALL
shape GPU-bound once its launch sequence is captured?

tools/host_floor.py says a small step is host-bound: 97-213 us of issue time per step (depending on how
loaded the box's host is) for 68-87 us of GPU work.  This tool captures one step (forward + backward through
the shim, the library's one-call passes) in a torch.cuda.CUDAGraph and times replays next to eager steps.

What capture needs from the op, and has: no device->host copy per call (level tables made once by
make_level_tables and known to the shim), every launch on torch's current stream (the shim passes
torch._C._cuda_getCurrentRawStream), workspaces from torch's caching allocator (graph-private pool under
capture), function attributes set during the eager warm-up.

Round 2 (profiles/r02ak_graph_step.jsonl): cfg1 150 -> 63 us, Nq = 64 96 -> 46 us, decode 96 -> 46 us per
forward + backward, replayed gradients equal to the eager ones.
usage: python tools/graph_step.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch  # noqa: E402
from mmfs_amd.functions import MSDeformAttnFunction  # noqa: E402
from mmfs_amd.levels import make_level_tables  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tools"))
from host_floor import SHAPES, make  # noqa: E402


def main():
    for name, w in SHAPES.items():
        value, sh, st, loc, attn, grad = make(w)

        def step():
            out = MSDeformAttnFunction.apply(value, sh, st, loc, attn, 64)
            return torch.autograd.grad(out, (value, loc, attn), grad)

        for _ in range(20):
            step()
        torch.cuda.synchronize()
        n = 500
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                  # (warm-up on the capture stream, as torch's notes ask)
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            captured = step()
        torch.cuda.synchronize()
        want = [g.clone() for g in step()]
        graph.replay()
        torch.cuda.synchronize()
        # (the order of a cell's records, hence of fp32 sums, is not fixed: compare with a tolerance)
        same = all(torch.allclose(a.float(), b.float(), rtol=1e-2, atol=1e-3) for a, b in zip(captured, want))
        for _ in range(20):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            graph.replay()
        torch.cuda.synchronize()
        replay = (time.perf_counter() - t0) / n
        print(json.dumps({"shape": name, "eager_us_per_step": round(eager * 1e6, 1),
                          "graph_us_per_step": round(replay * 1e6, 1), "replay_equals_eager": bool(same)}), flush=True)


if __name__ == "__main__":
    main()
