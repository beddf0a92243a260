#!/bin/bash
# GPU box: builder parameters (leaf size, cost of a node visit in triangle tests) on two scenes
cd "$GRAFT_REPO_ROOT"
for cfg in "4 4.0" "4 2.0" "4 1.0" "3 2.0" "2 2.0" "2 1.0" "6 4.0"; do
  set -- $cfg
  export RT355_LEAF_MAX=$1 RT355_TRAV_COST=$2
  echo "== leaf_max $1 trav_cost $2"
  bash scripts/bench_scene.sh north_star ns_l$1_t$2 --steps 64 --warmup 16 | head -1 || exit 1
  bash scripts/bench_scene.sh stress c5_l$1_t$2 --width 7680 --height 4320 --steps 4 --warmup 2 | head -1 || exit 1
done
