#!/bin/bash
# Runs on the GPU box (via gpurun): per-kernel time stats + PMC counter passes for bench.py.
# Usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the default command (four GOP pipelines: kernels of neighbouring steps overlap, so per-kernel durations
# include the co-runner) ...
BENCH2="python $ROOT/bench.py --no-cpu-baseline --no-verify $*"      # exactly the default command, minus the CPU leg (callers add --no-clips)
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_default" -o stats -- $BENCH2 > "$OUT/stats_default.log" 2>&1
# ... and one pipeline alone: every kernel has the chip to itself (this is what the counters describe)
BENCH="python $ROOT/bench.py --streams 1 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-verify $*"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats" -o stats -- $BENCH > "$OUT/stats.log" 2>&1
# ... and the same without the sole holder's rewrite of the shared hash table (force bit 15): the condition of bench.py's own
# "alone" figures, which are taken while the other pipelines' contexts still hold the table
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_norewrite" -o stats -- $BENCH --force-bits 32768 > "$OUT/stats_norewrite.log" 2>&1
PMC1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY"
PMC2="SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
rocprofv3 --output-format csv --pmc $PMC1 --kernel-trace -d "$OUT/pmc1" -o pmc -- $BENCH > "$OUT/pmc1.log" 2>&1
rocprofv3 --output-format csv --pmc $PMC2 --kernel-trace -d "$OUT/pmc2" -o pmc -- $BENCH > "$OUT/pmc2.log" 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc3" -o pmc -- $BENCH > "$OUT/pmc3.log" 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc4" -o pmc -- $BENCH > "$OUT/pmc4.log" 2>&1
# the query kernel WITHOUT its rewrite of the pixel-index hash table (what it does whenever several contexts share the table, i.e. in
# the default four-pipeline run; one pipeline alone rewrites it to keep it cached): force bit 15
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc5" -o pmc -- $BENCH --force-bits 32768 > "$OUT/pmc5.log" 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc6" -o pmc -- $BENCH --force-bits 32768 > "$OUT/pmc6.log" 2>&1
find "$OUT" -name "*.csv" | head -30
python $ROOT/tools/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
