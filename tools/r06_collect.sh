#!/bin/bash
# Round-6 evidence run on the GPU box (gpurun): everything DESIGN.md cites beyond tools/r06_profile.sh and tools/r06_shard_sweep.sh, written under gpurun_out/r06final/
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06final; mkdir -p $O
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err
for bits in 8 16; do for g in 1 4; do python tools/step_without_mask.py $bits $g 4 2>&1 | tail -1; done; done > $O/step_without_mask.txt 2>&1
bash tools/density_sweep.sh > $O/density_sweep.txt 2>&1
bash tools/large_frames.sh > $O/large_frames.txt 2>&1
for st in 1 2 3 4 6; do python bench.py --no-cpu-baseline --no-clips --no-legs --no-verify --streams $st 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.readline());print('streams %d: %.0f Mpixel/s median 20-step region, %.0f steady, alone %s' % ($st, d['value'], d['steady_state']['value'], d['kernels_ms_per_step_alone']))"; done > $O/streams_sweep.txt
bash tools/tile_sweep.sh > $O/config4_2160p_lds_tile_sweep.txt 2>&1
python tools/decode_bench.py > $O/decode_bench.txt 2>&1; python tools/decode_bench.py 3840 2160 9 >> $O/decode_bench.txt 2>&1
for l in 1 2 3; do E2E_BRIEF=1 python tools/e2e_leg.py 0 $l 2>&1 | tail -2; done > $O/e2e_lanes.txt 2>&1
( time python -m pytest tests -m gpu -q ) > $O/gpu_tests.txt 2>&1
timeout 200 python tools/fuzz_soak.py 120 > $O/fuzz_soak.txt 2>&1
timeout 150 python tools/fuzz_surface.py 90 > $O/fuzz_surface.txt 2>&1
tail -qn 3 $O/gpu_tests.txt $O/fuzz_soak.txt $O/fuzz_surface.txt $O/density_sweep.txt $O/step_without_mask.txt
bash tools/r06_profile.sh > $O/profile.log 2>&1; tail -5 $O/profile.log
