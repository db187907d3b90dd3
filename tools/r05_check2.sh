#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_check2; mkdir -p $O
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
