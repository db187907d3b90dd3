# Round-4 evidence run on the GPU box (gpurun): everything DESIGN.md cites, written under gpurun_out/r04final/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04final; mkdir -p $O
NL="--no-cpu-baseline --no-clips --no-legs"
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py $NL --interleaved 2>/dev/null | tail -1 > $O/bench_interleaved.json
python bench.py $NL --bits 16 2>/dev/null | tail -1 > $O/bench_uint16.json
python bench.py $NL --rebuild-hash-table 2>/dev/null | tail -1 > $O/bench_rebuild_hash_table.json
for st in 1 2 3 4 6; do python bench.py $NL --no-verify --streams $st 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('streams %d: %.0f Mpixel/s, %.4f ms/step, alone %s' % ($st, d['value'], d['ms_per_step'], d['kernels_ms_per_step_alone']))"; done > $O/streams_sweep.txt
for args in "--width 2560 --height 1440 --frames 30 --steps 60" "--width 3840 --height 2160 --frames 9 --steps 40" "--width 5120 --height 2880 --frames 9 --steps 20" "--width 7680 --height 4320 --frames 5 --steps 10"; do
python bench.py $NL $args 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['config']['workload'][:44], '| %.0f Mpixel/s | %.4f ms/step | alone' % (d['value'], d['ms_per_step']), d['kernels_ms_per_step_alone'], '| verified', d.get('verified_vs_oracle',{}).get('frames'))"; done > $O/large_frames.txt
for p in 0.01 0.05 0.2 0.3; do python bench.py $NL --density $p 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('p=$p | %.0f Mpixel/s | %.4f ms/step | alone' % (d['value'], d['ms_per_step']), d['kernels_ms_per_step_alone'], '| verified', d.get('verified_vs_oracle',{}).get('frames'))"; done > $O/density_sweep.txt
bash tools/profile.sh r04final --no-clips --no-legs > $O/profile.log 2>&1
cp gpurun_out/prof_r04final/summary.txt $O/rocprofv3_summary.txt
cp gpurun_out/prof_r04final/stats/*kernel_stats.csv $O/kernel_stats_streams1.csv 2>/dev/null
cp gpurun_out/prof_r04final/stats_norewrite/*kernel_stats.csv $O/kernel_stats_streams1_norewrite.csv 2>/dev/null
cp gpurun_out/prof_r04final/stats_default/*kernel_stats.csv $O/kernel_stats_default_4pipelines.csv 2>/dev/null
bash tools/profile.sh r04final_2160p --no-clips --no-legs --width 3840 --height 2160 --frames 9 > $O/profile_2160p.log 2>&1
cp gpurun_out/prof_r04final_2160p/summary.txt $O/rocprofv3_summary_2160p.txt
cp gpurun_out/prof_r04final_2160p/stats/*kernel_stats.csv $O/kernel_stats_2160p_streams1.csv 2>/dev/null
bash tools/tile_sweep.sh > $O/config4_2160p_lds_tile_sweep.txt 2>&1
bash tools/r04_trace.sh r04final/overlap_tmp > /dev/null 2>&1; cp gpurun_out/r04final/overlap_tmp/overlap.txt $O/overlap_4pipelines.txt 2>/dev/null
bash tools/r04_sensitivity.sh > /dev/null 2>&1; cp gpurun_out/r04i/sensitivity.txt $O/step_sensitivity.txt 2>/dev/null
python tools/decode_bench.py > $O/decode_bench.txt 2>&1; python tools/decode_bench.py 3840 2160 9 >> $O/decode_bench.txt 2>&1
./build/bench_query5 2 > $O/query_u64_harness.txt 2>&1
python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1
timeout 400 python tools/fuzz_soak.py 240 > $O/fuzz_soak.txt 2>&1
ls -la $O
