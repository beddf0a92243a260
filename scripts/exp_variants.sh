#!/bin/bash
# Headline bench under several settings, one line each (run on the GPU box).
# usage: scripts/exp_variants.sh "name|ENV=1 ENV2=2|--bench-arg v ..." ...
out=gpurun_out/variants.txt
: > $out
for spec in "$@"; do
  IFS='|' read -r name envs bargs <<< "$spec"
  env $envs timeout -k 10 150 python3 bench.py --no-cpu-baseline --no-extra --steps 48 --warmup 12 $bargs > gpurun_out/v.log 2>&1 || { echo "$name FAILED" >> $out; tail -3 gpurun_out/v.log >> $out; break; }
  python3 - "$name" >> $out <<'PY'
import json,sys
l=[x for x in open('gpurun_out/v.log') if x.startswith('{')][-1]
b=json.loads(l)
c=b['config']
r=b['roofline']
agg={}
for x in r['launches']: agg[x['kernel'][2:]]=agg.get(x['kernel'][2:],0)+x['ms']
print(f"{sys.argv[1]:24s} K={c['chains_in_flight']} B={c['frames_per_chain']} ms/step {b['ms_per_step']:.4f}  single {c['single_frame_ms']:.4f}  Mrays/s {b['value']:.0f}  chain-sum/frame {r['frame_kernel_ms']:.4f}  " + " ".join(f"{k}:{v:.3f}" for k,v in agg.items()))
PY
done
cat $out
