#!/bin/bash
# GPU box: generations traced in packets (the rest one lane per ray), headline scene, throughput chains
cd "$GRAFT_REPO_ROOT"
for pg in 1 2 3 6; do
  RT355_PACKET_GENS=$pg bash scripts/bench_scene.sh north_star ns_pg$pg --steps 96 --warmup 16 || exit 1
done
