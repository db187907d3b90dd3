cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r02_tests_j.txt 2>&1; tail -3 gpurun_out/r02_tests_j.txt
python -c "import __graft_entry__ as g; g.smoke()"
bash tools/collect_r02.sh > /dev/null 2>&1; cat gpurun_out/r02final/streams_sweep.txt; head -30 gpurun_out/r02final/bench_query.txt
