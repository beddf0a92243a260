#!/bin/bash
# Short bench of another scene (GPU box): scripts/bench_scene.sh <scene> [tag] [bench args]; environment passes through
cd "$GRAFT_REPO_ROOT"
s=$1; tag=${2:-$1}; shift; shift
timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-extra --steps 24 --warmup 4 --scene "$s" "$@" > gpurun_out/bench_$tag.log 2>&1 || { tail -5 gpurun_out/bench_$tag.log; exit 1; }
python3 - "$tag" <<'PY'
import json, sys
v = sys.argv[1]
l = [x for x in open(f'gpurun_out/bench_{v}.log') if x.startswith('{')][-1]
d = json.loads(l)
print(v, "ms/step", d["ms_per_step"], "single", d["config"]["single_frame_ms"], "chain-sum", d["roofline"]["frame_kernel_ms"])
print("   " + " ".join(f'{x["kernel"][2:6]}{x["gen"]}:{x["ms"]:.3f}' for x in d["roofline"]["launches"]))
PY
