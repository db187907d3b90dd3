# Round-3 evidence run on the GPU box (gpurun): everything DESIGN.md cites, written under gpurun_out/r03final/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03final; mkdir -p $O
./build/bench_query3 2 > $O/bench_query3.txt 2>&1
./build/bench_query4 > $O/bench_query4.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --no-clips --interleaved 2>/dev/null | tail -1 > $O/bench_interleaved.json
python bench.py --no-cpu-baseline --no-clips --bits 16 2>/dev/null | tail -1 > $O/bench_uint16.json
python bench.py --no-cpu-baseline --no-clips --rebuild-hash-table 2>/dev/null | tail -1 > $O/bench_rebuild_hash_table.json
for st in 1 2 3 4 6; do python bench.py --no-cpu-baseline --no-verify --no-clips --streams $st 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('streams %d: %.0f Mpixel/s, %.4f ms/step, alone %s' % ($st, d['value'], d['ms_per_step'], d['kernels_ms_per_step_alone']))"; done > $O/streams_sweep.txt
for args in "--width 2560 --height 1440 --frames 30 --steps 60" "--width 3840 --height 2160 --frames 9 --steps 40" "--width 5120 --height 2880 --frames 9 --steps 20" "--width 7680 --height 4320 --frames 5 --steps 10"; do
python bench.py --no-cpu-baseline --no-clips $args 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['config']['workload'][:44], '| %.0f Mpixel/s | %.4f ms/step | alone' % (d['value'], d['ms_per_step']), d['kernels_ms_per_step_alone'], '| verified', d.get('verified_vs_oracle',{}).get('frames'))"; done > $O/large_frames.txt
for p in 0.01 0.05 0.2 0.3; do python bench.py --no-cpu-baseline --no-clips --density $p 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('p=$p | %.0f Mpixel/s | %.4f ms/step | alone' % (d['value'], d['ms_per_step']), d['kernels_ms_per_step_alone'], '| verified', d.get('verified_vs_oracle',{}).get('frames'))"; done > $O/density_sweep.txt
bash tools/profile.sh r03final --no-clips > $O/profile.log 2>&1
cp gpurun_out/prof_r03final/summary.txt $O/rocprofv3_summary.txt
cp gpurun_out/prof_r03final/stats/*kernel_stats.csv $O/kernel_stats_streams1.csv 2>/dev/null
cp gpurun_out/prof_r03final/stats_norewrite/*kernel_stats.csv $O/kernel_stats_streams1_norewrite.csv 2>/dev/null
cp gpurun_out/prof_r03final/stats_default/*kernel_stats.csv $O/kernel_stats_default_4pipelines.csv 2>/dev/null
bash tools/profile.sh r03final_2160p --no-clips --width 3840 --height 2160 --frames 9 > $O/profile_2160p.log 2>&1
cp gpurun_out/prof_r03final_2160p/summary.txt $O/rocprofv3_summary_2160p.txt
cp gpurun_out/prof_r03final_2160p/stats/*kernel_stats.csv $O/kernel_stats_2160p_streams1.csv 2>/dev/null
bash tools/tile_sweep.sh > $O/config4_2160p_lds_tile_sweep.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ktrace && rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-verify --no-kernel-timing --no-clips --steps 200 --exact-steps > /tmp/ktrace.log 2>&1 )
python tools/overlap_report.py $(find /tmp/ktrace -name "*kernel_trace.csv" | head -1) > $O/overlap_4pipelines.txt 2>&1
python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1
ls -la $O
