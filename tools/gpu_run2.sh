set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./build/opbench > gpurun_out/r02_opbench2.txt 2>&1; tail -22 gpurun_out/r02_opbench2.txt
(cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt 2>&1)
bash tools/pmc_query.sh r02a > gpurun_out/r02_pmc_query_a.txt 2>&1; tail -80 gpurun_out/r02_pmc_query_a.txt
timeout 900 python -m pytest tests/test_gpu_dist_nccl.py -x -q -m gpu > gpurun_out/r02_tests_b.txt 2>&1; tail -15 gpurun_out/r02_tests_b.txt
