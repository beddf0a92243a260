"""Aggregates rocprofv3 --pmc CSVs (one pass per file) into a per-kernel table."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void rt355::", "")
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]] += 1
for k in sorted(agg):
    print(f"== {k}")
    for c in sorted(agg[k]):
        n = calls[k][c]
        print(f"   {c:28s} total {agg[k][c]:.6g}   per launch {agg[k][c] / n:.6g}   ({n} launches)")
    a = agg[k]
    n = max(1, calls[k].get("SQ_WAVES", 1))
    if "SQ_INSTS_VALU" in a and "SQ_THREAD_CYCLES_VALU" in a and a.get("SQ_ACTIVE_INST_VALU"):
        # THREAD_CYCLES_VALU / (ACTIVE_INST_VALU * 64): fraction of lanes active in issued VALU instructions
        print(f"   -> VALU lane utilisation ~ {a['SQ_THREAD_CYCLES_VALU'] / (a['SQ_ACTIVE_INST_VALU'] * 64):.3f}")
    if a.get("SQ_WAVE_CYCLES"):
        print(f"   -> wave-cycle split: wait_any {a.get('SQ_WAIT_ANY', 0) / a['SQ_WAVE_CYCLES']:.3f} wait_inst {a.get('SQ_WAIT_INST_ANY', 0) / a['SQ_WAVE_CYCLES']:.3f} valu_active {a.get('SQ_ACTIVE_INST_VALU', 0) / a['SQ_WAVE_CYCLES']:.3f}")
    if a.get("TCC_HIT_sum") is not None and (a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 0)) > 0:
        print(f"   -> L2 hit rate {a['TCC_HIT_sum'] / (a['TCC_HIT_sum'] + a['TCC_MISS_sum']):.4f}")
    f64 = a.get("SQ_INSTS_VALU_ADD_F64", 0) + a.get("SQ_INSTS_VALU_MUL_F64", 0) + a.get("SQ_INSTS_VALU_TRANS_F64", 0) + 2 * a.get("SQ_INSTS_VALU_FMA_F64", 0)
    if f64:
        print(f"   -> FP64 wave-instructions x64 lanes (upper bound on executed FLOPs): {f64 * 64:.6g}")
