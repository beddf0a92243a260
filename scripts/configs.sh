#!/bin/bash
# One short bench line per benchmark configuration (run on the GPU box).  usage: scripts/configs.sh [real_bits]
cd "$GRAFT_REPO_ROOT"
real=${1:-64}
run() {
  echo "== $*"
  timeout -k 10 280 python -u bench.py --no-cpu-baseline --real $real "$@" > gpurun_out/cfg.log 2>&1 || { tail -5 gpurun_out/cfg.log; exit 1; }
  python3 scripts/bsum.py gpurun_out/cfg.log
  grep '^{' gpurun_out/cfg.log >> gpurun_out/configs_$real.jsonl
}
rm -f gpurun_out/configs_$real.jsonl
run --scene ch9 --width 1920 --height 1080 --steps 96 &&
run --scene teapot_low --width 1920 --height 1080 --steps 96 &&
run --scene teapot_high --width 3840 --height 2160 --steps 24 &&
run --scene glass --width 1920 --height 1080 --steps 96 &&
run --scene north_star --width 1920 --height 1080 --steps 96 &&
run --scene north_star --width 3840 --height 2160 --steps 24 &&
run --scene stress --width 7680 --height 4320 --steps 6 --warmup 2 &&
run --scene teapot_field --width 1920 --height 1080 --steps 48
