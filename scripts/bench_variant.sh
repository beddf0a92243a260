#!/bin/bash
# Short bench of one or more library variants (GPU box): scripts/bench_variant.sh name1 name2 ...  (build_exp/lib<name>.so; "base" = in-tree build)
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" = base ]; then unset RT355_LIB; else export RT355_LIB=build_exp/lib$v.so; fi
  timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-extra --steps 48 --warmup 8 > gpurun_out/bench_$v.log 2>&1 || { tail -5 gpurun_out/bench_$v.log; exit 1; }
  python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1]
l = [x for x in open(f'gpurun_out/bench_{v}.log') if x.startswith('{')][-1]
d = json.loads(l)
print(v, "ms/step", d["ms_per_step"], "single", d["config"]["single_frame_ms"], "chain-sum", d["roofline"]["frame_kernel_ms"])
print("   " + " ".join(f'{x["kernel"][2:6]}{x["gen"]}:{x["ms"]:.3f}' for x in d["roofline"]["launches"]))
PY
done
