#!/bin/bash
# Copies the summaries of one scripts/profile.sh run into profiles/: scripts/collect_profiles.sh <tag> <round prefix, e.g. r03>
tag=$1; r=$2
src=gpurun_out/prof_$tag
cd "$(dirname "$0")/.."
python3 scripts/make_counters.py $src/pmc $src/counters.json 16 || exit 1
cp $src/bench.json profiles/${r}_bench_under_rocprof.json
cp $src/bench_k3.json profiles/${r}_bench_default_run_under_rocprof.json
cp $src/kernel_stats.csv profiles/${r}_kernel_stats.csv
cp $src/kernel_stats_k3.csv profiles/${r}_kernel_stats_default_run.csv
cp $src/kernel_stats_by_generation.txt profiles/${r}_kernel_stats_by_generation.txt
cp $src/counters.json profiles/${r}_counters.json
cp $src/pmc_per_dispatch.txt profiles/${r}_pmc_per_dispatch.txt
cp $src/pmc_summary.txt profiles/${r}_pmc_summary.txt
cp $src/timeline.txt profiles/${r}_timeline.txt
ls -la profiles/${r}_*
