#!/bin/bash
# The headline bench at the driver's run length (20 steps, 5 warm-up) for several chain shapes (run on the GPU box).
out=gpurun_out/steps.txt
: > $out
for spec in "$@"; do
  IFS='|' read -r name bargs <<< "$spec"
  for rep in 1 2 3; do
    timeout -k 10 150 python3 bench.py --no-cpu-baseline --no-extra $bargs > gpurun_out/v.log 2>&1 || { echo "$name FAILED" >> $out; tail -3 gpurun_out/v.log >> $out; break 2; }
    python3 - "$name" >> $out <<'PY'
import json,sys
l=[x for x in open('gpurun_out/v.log') if x.startswith('{')][-1]
b=json.loads(l); c=b['config']
print(f"{sys.argv[1]:28s} K={c['chains_in_flight']} B={c['frames_per_chain']} steps {b['steps']} ms/step {b['ms_per_step']:.4f} Mrays/s {b['value']:.0f}")
PY
  done
done
cat $out
