#!/bin/bash
# Copy what tools/collect_r04.sh wrote under gpurun_out/r04final/ into profiles/r04_* (the tracked, judged copies) and rebuild the
# derived files (issue model, query traffic, kernel resources).  Runs in the build container (no GPU).
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r04final
for f in bench_default.json bench_interleaved.json bench_uint16.json bench_rebuild_hash_table.json streams_sweep.txt large_frames.txt density_sweep.txt \
         rocprofv3_summary.txt rocprofv3_summary_2160p.txt kernel_stats_default_4pipelines.csv kernel_stats_streams1.csv kernel_stats_streams1_norewrite.csv \
         kernel_stats_2160p_streams1.csv config4_2160p_lds_tile_sweep.txt overlap_4pipelines.txt step_sensitivity.txt decode_bench.txt query_u64_harness.txt gpu_tests.txt fuzz_soak.txt; do
    [ -f $S/$f ] && cp $S/$f profiles/r04_$f
done
python tools/make_r04_models.py profiles/r04_rocprofv3_summary.txt
python tools/kernel_resources.py > profiles/r04_kernel_resources.txt
ls profiles/r04_*
