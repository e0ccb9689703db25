#!/usr/bin/env python3
"""Per-phase timing of msda_bwd_value_tiled (experimental build with -DMMFS_VAL_TIMING=<u32 offset>).
The kernel stamps clock64() at phase boundaries into the cursor area + offset; this script calls the
C ABI directly with an over-sized workspace and prints the per-phase mean over workgroups."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.environ["MMFS_MSDA_LIB"])
i64, vp = ctypes.c_int64, ctypes.c_void_p
lib.mmfs_msda_backward_value.restype = ctypes.c_int
lib.mmfs_msda_backward_value.argtypes = [ctypes.c_int] + [vp] * 7 + [i64] * 8 + [vp]
lib.mmfs_msda_backward_workspace_bytes.restype = i64
lib.mmfs_msda_backward_workspace_bytes.argtypes = [ctypes.c_int] + [i64] * 7 + [ctypes.c_uint]
B, Nq, H, D, P = 8, 4096, 8, 128, 4
shapes = torch.tensor([(64, 64), (32, 32), (16, 16), (8, 8)], device="cuda")
start = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
S, L = int(shapes.prod(1).sum()), 4
dt = torch.bfloat16
loc = torch.rand(B, Nq, H, L, P, 2, device="cuda").to(dt)
attn = torch.rand(B, Nq, H, L, P, device="cuda").to(dt)
go = torch.randn(B, Nq, H * D, device="cuda").to(dt)
gv = torch.empty(B, S, H, D, device="cuda", dtype=dt)
dims = (B, S, H, D, L, Nq, P)
need = lib.mmfs_msda_backward_workspace_bytes(2, *dims, 1)
OFF = int(os.environ.get("STAMP_OFF", 1 << 16))          # in uint32 units from the cursor array
ws = torch.zeros(need + 64 * 1024 * 1024, dtype=torch.uint8, device="cuda")
for _ in range(3):
    rc = lib.mmfs_msda_backward_value(2, shapes.data_ptr(), start.data_ptr(), loc.data_ptr(), attn.data_ptr(),
                                      go.data_ptr(), gv.data_ptr(), ws.data_ptr(), ws.numel(), *dims, None)
    assert rc == 0, rc
torch.cuda.synchronize()
pts = B * Nq * H * L * P
up = lambda v: (v + 15) // 16 * 16
cursor_off = up(pts * 2 * 2) + up(pts * 2)
st = ws[cursor_off + OFF * 4:].view(torch.int64)[: 8 * 20000].view(-1, 8).cpu()
st = st[st[:, 0] != 0]
print("workgroups with stamps:", st.shape[0])
d = (st[:, 1:6] - st[:, 0:5]).double()
names = ["setup+zero", "count scan", "prefix", "alloc+scatter scan", "reduce"]
for i, n in enumerate(names):
    print(f"  {n:20s} mean {d[:, i].mean() / 1e3:8.1f} kclk   max {d[:, i].max() / 1e3:8.1f}")
print("  total per WG mean kclk", (st[:, 5] - st[:, 0]).double().mean().item() / 1e3, " records/WG", st[:, 6].double().mean().item(), "px/WG", st[:, 7].double().mean().item())
for npx in sorted(set(st[:, 7].tolist())):
    m = st[:, 7] == npx
    dd = d[m]
    print(f"  tiles with {int(npx):5d} px: n={int(m.sum()):4d} records/WG {st[m, 6].double().mean():9.0f}  "
          + "  ".join(f"{n.split()[0]} {dd[:, i].mean() / 1e3:7.1f}" for i, n in enumerate(names)))
span = (st[:, 5].max() - st[:, 0].min()).item()
print("  first start -> last end:", span / 1e3, "kclk")
