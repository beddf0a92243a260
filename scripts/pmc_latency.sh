#!/bin/bash
# Two PMC passes over one launch chain: scalar-cache and instruction-cache hit rates, and the average time a scalar /
# vector memory instruction and an instruction fetch stay outstanding (LEVEL counters accumulate the number in flight
# per cycle: LEVEL / count = cycles each).  usage: scripts/pmc_latency.sh <outdir> [prof_frame.py args...]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
timeout -s KILL 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d "$out" -o c1 -- python3 scripts/prof_frame.py "$@" > "$out/c1.log" 2>&1 || { echo "pass 1 failed"; tail -3 "$out/c1.log"; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_LDS SQ_INST_LEVEL_LDS --output-format csv -d "$out" -o c2 -- python3 scripts/prof_frame.py "$@" > "$out/c2.log" 2>&1 || { echo "pass 2 failed"; tail -3 "$out/c2.log"; exit 1; }
python3 scripts/pmc_rows.py "$out" k_ SQ_WAVES SQ_WAVE_CYCLES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_LDS SQ_INST_LEVEL_LDS > "$out/rows.txt"
cat "$out/rows.txt"
