#!/bin/bash
# Kernel timeline of a few frames (run on the GPU box).  usage: scripts/ktrace.sh <outdir> [prof_frame.py args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o kt -- python3 scripts/prof_frame.py "$@" > "$out/kt.log" 2>&1 || { tail -5 "$out/kt.log"; exit 1; }
f=$(find "$out" -name 'kt_kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last frame: from the last k_reset_ctl on
idx = max(i for i, r in enumerate(rows) if "k_reset_ctl" in r["Kernel_Name"])
fr = rows[idx:]
t0 = int(fr[0]["Start_Timestamp"])
for r in fr:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void rt355::", "")
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{name:34s} start {s/1e3:8.1f}  end {e/1e3:8.1f}  dur {(e-s)/1e3:7.1f} us  stream/queue {r.get('Queue_Id','?')}")
PY
