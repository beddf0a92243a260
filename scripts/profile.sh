#!/bin/bash
# Profiles of the headline bench (run on the GPU box): kernel-trace statistics of `bench.py`, then the PMC
# passes over one frame.  usage: scripts/profile.sh <tag>   -> gpurun_out/prof_<tag>/
tag=${1:-r02}
out=gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
# one chain (16 frames) in flight at a time: each kernel's duration is its own (chains in flight overlap each other's kernels)
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o bench -- python3 bench.py --no-cpu-baseline --no-extra --in-flight 1 --steps 48 --warmup 16 > "$out/bench.log" 2>&1 || { tail -5 "$out/bench.log"; exit 1; }
grep '^{' "$out/bench.log" | tail -1 > "$out/bench.json"
f=$(find "$out" -name 'bench_kernel_stats.csv' | head -1)
cp "$f" "$out/kernel_stats.csv" && head -12 "$out/kernel_stats.csv"
python3 scripts/ktrace_by_generation.py "$(find "$out" -maxdepth 3 -name 'bench_kernel_trace.csv' | head -1)" > "$out/kernel_stats_by_generation.txt"
# the default command (several frames per chain, three chains in flight): the same kernels over larger generations, overlapped
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/k3" -o bench -- python3 bench.py --no-cpu-baseline --no-extra --steps 144 --warmup 48 > "$out/bench_k3.log" 2>&1 || { tail -5 "$out/bench_k3.log"; exit 1; }
grep '^{' "$out/bench_k3.log" | tail -1 > "$out/bench_k3.json"
cp "$(find "$out/k3" -name 'bench_kernel_stats.csv' | head -1)" "$out/kernel_stats_k3.csv"
bash scripts/pmc.sh "$out/pmc" --frames 1 || exit 1
python3 scripts/pmc_summary.py "$out/pmc" > "$out/pmc_summary.txt"
python3 scripts/pmc_rows.py "$out/pmc" k_ SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE > "$out/pmc_per_dispatch.txt"
python3 scripts/make_counters.py "$out/pmc" "$out/counters.json" 16 || exit 1
bash scripts/ktrace.sh "$out/kt" --frames 3 > "$out/timeline.txt" 2>&1
echo "profile done"
