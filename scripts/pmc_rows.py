"""Per-dispatch view of the rocprofv3 --pmc CSVs: one row per dispatch of the kernels matching argv[2]."""
import csv, glob, os, re, sys
from collections import defaultdict
d, pat = sys.argv[1], sys.argv[2]
rows = defaultdict(dict)
names = {}
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    tag = os.path.basename(f).split("_counter")[0]
    # dispatch ids restart per pass; order of dispatches is identical across passes
    per = defaultdict(dict)
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void rt355::", "")
        per[int(row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
        names[(tag, int(row["Dispatch_Id"]))] = k
    for i, did in enumerate(sorted(per)):
        rows[i].update(per[did])
        rows[i]["_name"] = names[(tag, did)]
cols = sys.argv[3:] or ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum"]
print("idx name " + " ".join(cols))
for i in sorted(rows):
    r = rows[i]
    if pat in r["_name"]:
        print(i, r["_name"], " ".join("%.4g" % r.get(c, float("nan")) for c in cols))
