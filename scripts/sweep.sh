#!/bin/bash
# usage: scripts/sweep.sh "ENV1=a ENV2=b" "ENV1=c" ...   one short bench per environment setting
cd "$GRAFT_REPO_ROOT"
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 120 python -u bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/sweep.log 2>&1 || { tail -5 gpurun_out/sweep.log; exit 1; }
  python3 scripts/bsum.py gpurun_out/sweep.log
done
