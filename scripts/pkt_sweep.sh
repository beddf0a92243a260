#!/bin/bash
# GPU box: packets against one lane per ray, per scene and image size
cd "$GRAFT_REPO_ROOT"
run() {  # scene tag args...
  local s=$1 t=$2; shift; shift
  for pg in 0 1 6; do
    RT355_PACKET_GENS=$pg bash scripts/bench_scene.sh $s ${t}_pg$pg "$@" | head -1 || exit 1
  done
}
run stress c5_1080 --steps 12 --warmup 2
run stress c5_4k --width 3840 --height 2160 --steps 8 --warmup 2
run teapot_high th_1080 --steps 12 --warmup 2
run teapot_high th_4k --width 3840 --height 2160 --steps 8 --warmup 2
run teapot_field tf_4k --width 3840 --height 2160 --steps 8 --warmup 2
run teapot_field tf_8k --width 7680 --height 4320 --steps 6 --warmup 2
run north_star ns_540 --width 960 --height 540 --steps 24 --warmup 4
