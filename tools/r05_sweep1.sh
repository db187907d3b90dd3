#!/bin/bash
# (ran at commit 30a3087: --side-compact, --skip-kernels and --insert-grouped were removed from bench.py and the library afterwards; results in profiles/r05_sweep1.txt)
# round 5, sweep 1: the multi-GOP block (rbf_encode_runs, 4 x 30 frames per call) -- insert launch shape, gather cache policy, mask chunks
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_sweep1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_runs.py -x -q > $O/pytest_runs.txt 2>&1; tail -3 $O/pytest_runs.txt
C="--no-clips --no-legs --no-cpu-baseline --steps 40 --gops-per-call 4 --gops-per-pipeline 1"
run() { name=$1; shift; timeout 300 python bench.py $C "$@" > $O/$name.json 2> $O/$name.err; }
run g4_default
run g4_grouped --insert-grouped
run g4_s2 --insert-slices 2
run g4_s4 --insert-slices 4
run g4_s16 --insert-slices 16
RBF_LIB_PATH=$PWD/build/librbf_nt.so run g4_nt
RBF_LIB_PATH=$PWD/build/librbf_nt.so run g4_nt_s4 --insert-slices 4
run g4_chunks8 --force-bits $((8<<8))
run g4_chunks16 --force-bits $((16<<8))
run g4_streams2 --streams 2
run g4_streams3 --streams 3
C="--no-clips --no-legs --no-cpu-baseline --steps 100"
run g1_default
RBF_LIB_PATH=$PWD/build/librbf_nt.so run g1_nt
run g1_side --side-compact
C="--no-clips --no-legs --no-cpu-baseline --steps 40 --gops-per-call 4 --gops-per-pipeline 1"
run g4_side --side-compact
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
