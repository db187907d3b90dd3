#!/bin/bash
# round 5: the sharded path at world 2 / 8 on one device, the default bench line with its new legs, the GPU tier
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_check1; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_dist_shared.py -x -q ) > $O/pytest_shared.txt 2>&1; tail -6 $O/pytest_shared.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
( time timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_dist_shared.py ) > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
