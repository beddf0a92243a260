#!/bin/bash
# GPU box: frames per chain x chains in flight on the headline scene
cd "$GRAFT_REPO_ROOT"
for cfg in "8 3" "12 3" "16 3" "16 2" "24 2" "32 2"; do
  set -- $cfg
  timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-extra --steps 192 --warmup 16 --batch $1 --in-flight $2 > gpurun_out/bench_b$1_k$2.log 2>&1 || { tail -5 gpurun_out/bench_b$1_k$2.log; exit 1; }
  python3 - "$1" "$2" <<'PY'
import json, sys
l = [x for x in open(f'gpurun_out/bench_b{sys.argv[1]}_k{sys.argv[2]}.log') if x.startswith('{')][-1]
d = json.loads(l)
print("batch", sys.argv[1], "in flight", sys.argv[2], "ms/step", d["ms_per_step"], "chain-sum/frame", d["roofline"]["frame_kernel_ms"])
PY
done
