#!/bin/bash
# usage: scripts/sweep_cfg.sh "<bench args>" "ENV=a" "ENV=b" ...
cd "$GRAFT_REPO_ROOT"
args=$1; shift
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python -u bench.py --no-cpu-baseline $args > gpurun_out/sweep.log 2>&1 || { tail -5 gpurun_out/sweep.log; exit 1; }
  python3 scripts/bsum.py gpurun_out/sweep.log | cut -c1-200
done
