#!/bin/bash
# PMC passes over scripts/prof_frame.py, one rocprofv3 run per counter set (run on the GPU box).
# usage: scripts/pmc.sh <outdir> [prof_frame.py args...]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
run() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$out" -o "$name" -- python3 scripts/prof_frame.py $EXTRA > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$out/$name.log"; return 1; }
  echo "pass $name ok"
}
EXTRA="$*"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY &&
run sq2 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM &&
run tcc1 TCC_HIT_sum TCC_MISS_sum &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run grbm GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES
