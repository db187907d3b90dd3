#!/bin/bash
# Round-6 rocprofv3 evidence (runs on the GPU box via gpurun): per-kernel time stats of the default command and of five launch shapes
# alone (one pipeline), the HBM traffic counters of the same shapes -- each PMC counter in its own pass with --kernel-trace only -- and the
# instruction counters tools/make_issue_model.py prices (g1, g4, u4).
#   g1   1920x1080, 30-frame GOP, one GOP per call           (the contract line's shape)
#   g4   1920x1080, 4 x 30 frames per call (rbf_encode_runs)  (the batched_gops leg)
#   c4   3840x2160, 30-frame GOP                              (BASELINE configs[3])
#   u1   1920x1080 16-bit, one GOP per call                   (BASELINE configs[4]'s single-GPU half)
#   u4   1920x1080 16-bit, 4 x 30 frames per call
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_profile
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
NL="--no-cpu-baseline --no-verify --no-clips --no-legs"
SHAPES=${SHAPES:-"g1 g4 c4 u1 u4"}
if [ -z "${SKIP_DEFAULT:-}" ]; then
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_default" -o stats -- python $ROOT/bench.py $NL --steps 20 --warmup 5 > "$OUT/stats_default.log" 2>&1
fi
ALONE="--streams 1 --steps 10 --warmup 2 --exact-steps --no-kernel-timing --force-bits 32768 $NL"
declare -A SHAPE=( [g1]="" [g4]="--gops-per-call 4 --gops-per-pipeline 1" [c4]="--width 3840 --height 2160 --frames 30 --gops-per-pipeline 1"
                   [u1]="--bits 16" [u4]="--bits 16 --gops-per-call 4 --gops-per-pipeline 1" )
declare -A STEPS=( [g1]=160 [g4]=48 [c4]=16 [u1]=160 [u4]=48 )      # (time stats over many launches: the first ones after start-up run at lower clocks)
for s in $SHAPES; do
  rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_$s" -o stats -- python $ROOT/bench.py $ALONE ${SHAPE[$s]} --steps ${STEPS[$s]} --warmup 8 > "$OUT/stats_$s.log" 2>&1
  rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch_$s" -o pmc -- python $ROOT/bench.py $ALONE ${SHAPE[$s]} > "$OUT/pmc_fetch_$s.log" 2>&1
  rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write_$s" -o pmc -- python $ROOT/bench.py $ALONE ${SHAPE[$s]} > "$OUT/pmc_write_$s.log" 2>&1
done
INST_SHAPES=${INST_SHAPES:-g1 g4 u4}
for s in $INST_SHAPES; do
  rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d "$OUT/pmc_insts_$s" -o pmc -- python $ROOT/bench.py $ALONE ${SHAPE[$s]} > "$OUT/pmc_insts_$s.log" 2>&1
  rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --kernel-trace -d "$OUT/pmc_insts2_$s" -o pmc -- python $ROOT/bench.py $ALONE ${SHAPE[$s]} > "$OUT/pmc_insts2_$s.log" 2>&1
done
python $ROOT/tools/make_traffic.py "$OUT" r06 > "$OUT/summary.txt" 2>&1
tail -120 "$OUT/summary.txt"
