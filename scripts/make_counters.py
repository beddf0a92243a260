"""rocprofv3 --pmc passes over one launch chain (scripts/pmc.sh) -> profiles-style JSON keyed by launch:
   python scripts/make_counters.py <pmc_dir> <out.json> [frames_per_chain]
HBM bytes follow the gfx950 convention for FETCH_SIZE / WRITE_SIZE (KiB; wide reads counted at half size, so
FETCH_SIZE x 2).  The file is stamped with a hash of the device sources; bench.py quotes it only when the hash
matches the tree."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
d, out_path = sys.argv[1], sys.argv[2]
rows = defaultdict(dict)
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    per, names = defaultdict(dict), {}
    for row in csv.DictReader(open(f)):
        did = int(row["Dispatch_Id"])
        per[did][row["Counter_Name"]] = float(row["Counter_Value"])
        names[did] = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void rt355::", "").replace("rt355::", "")
    for i, did in enumerate(sorted(per)):  # dispatch ids restart per pass; the order of dispatches is the same
        rows[i].update(per[did])
        rows[i]["_name"] = names[did]
seen = defaultdict(int)
launches = {}
for i in sorted(rows):
    r = rows[i]
    base = r["_name"].split("<")[0]
    if base not in ("k_trace", "k_shadow", "k_spawn", "k_combine", "k_scan", "k_trace_spawn"):
        continue
    gen = seen[base]
    seen[base] += 1
    g = lambda k: r.get(k, 0.0)
    fp64 = 64.0 * (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_TRANS_F64") + 2.0 * g("SQ_INSTS_VALU_FMA_F64"))
    e = {"kernel": r["_name"], "fetch_kib": g("FETCH_SIZE"), "write_kib": g("WRITE_SIZE"),
         "hbm_bytes": int((2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024), "fp64_flops_upper_bound": fp64}
    if g("SQ_ACTIVE_INST_VALU"):
        e["lane_utilisation"] = round(g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU")), 4)
    if g("SQ_WAVE_CYCLES"):
        e["wait_any_frac"] = round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 4)
        e["valu_active_frac"] = round(g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), 4)
    if g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
        e["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
    e["waves"] = int(g("SQ_WAVES"))
    e["valu_instructions"] = int(g("SQ_INSTS_VALU"))
    # the combine launches run from the deepest folded generation back to generation 0: keyed by their order
    launches[f"{base} launch {gen}" if base == "k_combine" else f"{base} generation {gen}"] = e
import bench
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 1
doc = {"source": f"rocprofv3 --pmc, one pass per counter set (scripts/pmc.sh), north_star 1920x1080 depth 5 f64, 1 GPU, one launch chain of {frames} frames; per launch",
       "frames_per_chain": frames,
       "correction": "hbm_bytes = (FETCH_SIZE x 2 + WRITE_SIZE) KiB: gfx950 counts wide reads at half size",
       "source_hash": bench.source_hash(),
       "per_frame": {"valu_instructions": int(sum(e["valu_instructions"] for e in launches.values()) / frames),
                     "hbm_bytes": int(sum(e["hbm_bytes"] for e in launches.values()) / frames),
                     "fp64_flops_upper_bound": sum(e["fp64_flops_upper_bound"] for e in launches.values()) / frames},
       "launches": launches}
json.dump(doc, open(out_path, "w"), indent=1)
print("wrote", out_path, len(launches), "launches")
