#!/bin/bash
# VALU instruction counts per launch of one frame for cut-down versions of a scene (run on the GPU box).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/fixed
mkdir -p $out
for keep in floor,matte floor floor,sphere floor,mesh,matte floor,mesh floor,sphere,mesh; do
  tag=$(echo $keep | tr ',' '_')
  timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/$tag -o p -- python3 scripts/prof_frame.py --frames 1 --keep $keep > $out/$tag.log 2>&1 || { echo "$tag failed"; tail -3 $out/$tag.log; exit 1; }
  echo "== $keep" >> $out/rows.txt
  python3 scripts/pmc_rows.py $out/$tag k_ SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE >> $out/rows.txt
done
echo done
