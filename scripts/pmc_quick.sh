#!/bin/bash
# One PMC pass (instruction and wait counters) over one launch chain; prints per-dispatch rows.
# usage: scripts/pmc_quick.sh <outdir> [prof_frame.py args...]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$out" -o sq1 -- python3 scripts/prof_frame.py "$@" > "$out/sq1.log" 2>&1 || { echo "pass failed"; tail -3 "$out/sq1.log"; exit 1; }
python3 scripts/pmc_rows.py "$out" k_ SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU > "$out/rows.txt"
tail -27 "$out/rows.txt"
