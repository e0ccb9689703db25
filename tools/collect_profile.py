#!/usr/bin/env python3
"""Copies the judged summaries of a tools/prof.sh run (gpurun_out/prof_<tag>/) into profiles/
and derives profiles/pmc_traffic.json (HBM bytes per launch per kernel, read by bench.py).

  python tools/collect_profile.py <tag> <round-prefix> [workload]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE shows half of a
wide coalesced read stream (MI355X_MICROARCH.md section HBM) -> doubled here, as the guide says.
"""
import csv, glob, json, os, re, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "cfg2_northstar"
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(src + "/trace/**/*kernel_stats.csv", recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(dst, f"{prefix}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        for r in rows[:16]:
            w.writerow(r)
shutil.copy(os.path.join(src, "pmc_summary.txt"), os.path.join(dst, f"{prefix}_pmc_summary.txt"))

names = {"msda_fwd_vec": "msda_fwd", "msda_bwd_value_reduce": "msda_bwd_value_reduce",
         "msda_bwd_value_sort": "msda_bwd_value_sort", "msda_bwd_vec_taps": "msda_bwd_taps",
         "msda_bwd_vec_atomic": "msda_bwd_atomic",
         # second-generation grad_value kernels answer to the same stage names in bench.py
         "msda_bwd_block_reduce": "msda_bwd_value_reduce", "msda_bwd_cell_sort": "msda_bwd_value_sort",
         # third generation: matrix-core tile reduce; the fused re-pack + plan launch
         "msda_bwd_tile_reduce": "msda_bwd_value_reduce", "msda_bwd_prepare": "msda_bwd_value_prepare",
         "msda_taps_coarse": "msda_bwd_taps_coarse",
         # round 3: LDS-resident levels on the matrix cores (forward; all-levels taps)
         "msda_fwd_mma": "msda_fwd", "msda_taps_mma": "msda_bwd_taps",
         # round 4: the sliced forward (heads of 32 / 64 channels, whole pyramid resident)
         "msda_fwd_q8": "msda_fwd",
         # round 5: a wave per query, weights on the diagonal of the A operand (the north star's forward)
         "msda_fwd_wq": "msda_fwd"}
traffic = {}
for line in open(os.path.join(src, "pmc_summary.txt")):
    kern, _, rest = line.partition(": ")
    vals = dict(kv.split("=") for kv in rest.strip().split(", ") if "=" in kv)
    if "FETCH_SIZE" not in vals:
        continue
    for pat, short in names.items():
        if pat in kern:
            rd = float(vals["FETCH_SIZE"]) * 1024 * 2          # gfx950 correction
            wr = float(vals.get("WRITE_SIZE", 0)) * 1024
            traffic[short] = int(rd + wr)
path = os.path.join(dst, "pmc_traffic.json")
allw = json.load(open(path)) if os.path.exists(path) else {}
allw[workload] = traffic
src_note = f"profiles/{prefix}_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; FETCH x2 per MI355X_MICROARCH.md)"
prov = allw.get("_source")
if not isinstance(prov, dict):          # (older files: one string for every workload)
    prov = {w: prov for w in allw if w != "_source"} if prov else {}
prov[workload] = src_note
allw["_source"] = prov
json.dump(allw, open(path, "w"), indent=1)
print(json.dumps(allw, indent=1))
