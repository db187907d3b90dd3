#!/bin/bash
# Round-5 rocprofv3 evidence (runs on the GPU box via gpurun): per-kernel time stats of the default command and of three launch shapes
# alone (one pipeline), and the HBM traffic counters of the same three shapes -- each PMC counter in its own pass with --kernel-trace only.
#   g1   1920x1080, 30-frame GOP, one GOP per call          (the contract line's shape)
#   g4   1920x1080, 4 x 30 frames per call (rbf_encode_runs) (the batched_gops leg)
#   c4   3840x2160, 30-frame GOP                             (BASELINE configs[3])
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_profile
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
NL="--no-cpu-baseline --no-verify --no-clips --no-legs"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_default" -o stats -- python $ROOT/bench.py $NL --steps 20 --warmup 5 > "$OUT/stats_default.log" 2>&1
ALONE="--streams 1 --steps 10 --warmup 2 --exact-steps --no-kernel-timing --force-bits 32768 $NL"
declare -A SHAPE=( [g1]="" [g4]="--gops-per-call 4 --gops-per-pipeline 1" [c4]="--width 3840 --height 2160 --frames 30 --gops-per-pipeline 1" )
declare -A STEPS=( [g1]=160 [g4]=48 [c4]=16 )      # (time stats over many launches: the first ones after start-up run at lower clocks)
for s in g1 g4 c4; do
  rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_$s" -o stats -- python $ROOT/bench.py $ALONE ${SHAPE[$s]} --steps ${STEPS[$s]} --warmup 8 > "$OUT/stats_$s.log" 2>&1
  rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch_$s" -o pmc -- python $ROOT/bench.py $ALONE ${SHAPE[$s]} > "$OUT/pmc_fetch_$s.log" 2>&1
  rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write_$s" -o pmc -- python $ROOT/bench.py $ALONE ${SHAPE[$s]} > "$OUT/pmc_write_$s.log" 2>&1
done
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d "$OUT/pmc_insts_g4" -o pmc -- python $ROOT/bench.py $ALONE ${SHAPE[g4]} > "$OUT/pmc_insts_g4.log" 2>&1
python $ROOT/tools/make_r05_traffic.py "$OUT" > "$OUT/summary.txt" 2>&1
tail -80 "$OUT/summary.txt"
