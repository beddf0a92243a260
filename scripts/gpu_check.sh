#!/bin/bash
# GPU box: parity tests, then the bench.  usage: scripts/gpu_check.sh [bench args]
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u bench.py --no-cpu-baseline "$@" > gpurun_out/bench.log 2>&1 || { tail -20 gpurun_out/bench.log; exit 1; }
python3 - <<'PY'
import json
l=[x for x in open('gpurun_out/bench.log') if x.startswith('{')][-1]
d=json.loads(l)
print("value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["kernel"], "frame_kernel_ms", d["roofline"]["frame_kernel_ms"])
print(" ".join(f'{x["kernel"][2:6]}{x["gen"]}:{x["ms"]:.3f}' for x in d["roofline"]["launches"]))
PY
