#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06c; mkdir -p $O
for bits in 8 16; do for g in 1 4; do python tools/step_without_mask.py $bits $g 4 2>&1 | tail -1; done; done | tee $O/step_without_mask.txt
python tools/step_without_mask.py 8 4 2 2>&1 | tail -1 | tee -a $O/step_without_mask.txt
python tools/step_without_mask.py 16 4 6 2>&1 | tail -1 | tee -a $O/step_without_mask.txt
for l in 2 3; do E2E_BRIEF=1 python tools/e2e_leg.py 0 $l 2>&1 | tail -2; done | tee $O/e2e.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace_u16_g4 -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-legs --no-clips --no-verify --no-kernel-timing --bits 16 --gops-per-call 4 --gops-per-pipeline 1 --steps 40 --warmup 8 > /dev/null 2>&1
rocprofv3 --output-format csv --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace_u8_g4 -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-legs --no-clips --no-verify --no-kernel-timing --bits 8 --gops-per-call 4 --gops-per-pipeline 1 --steps 40 --warmup 8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for t in u16_g4 u8_g4; do echo "== $t"; python tools/overlap_report.py $(find $O/trace_$t -name "*kernel_trace.csv" | head -1) 2>&1 | head -24; done | tee $O/overlap.txt
python -m pytest tests/test_gpu_dist_shared.py tests/test_gpu_dist_nccl.py tests/test_gpu_surface.py -q -x 2>&1 | tail -5
