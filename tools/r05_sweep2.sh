#!/bin/bash
# (ran at commit 30a3087: --skip-kernels was removed from bench.py and the library afterwards; results in profiles/r05_sweep2.txt)
# round 5, sweep 2: insert slices x pipelines on the 4-GOP block; what every kernel costs the overlapped step (results wrong with --skip-kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_sweep2; mkdir -p $O
C="--no-clips --no-legs --no-cpu-baseline --steps 40 --gops-per-call 4 --gops-per-pipeline 1"
run() { name=$1; shift; timeout 300 python bench.py $C "$@" > $O/$name.json 2> $O/$name.err; }
for s in 1 2 3; do for p in 2 3 4; do run g4_s${s}_p${p} --insert-slices $s --streams $p; done; done
run g4_s2_ahead1 --insert-slices 2 --begin-ahead 1
run g4_s2_ahead3 --insert-slices 2 --begin-ahead 3
run g4_s2_threads --insert-slices 2 --host-threads 1
for k in stitch reduce insert query "insert,reduce,stitch" "insert,reduce,query,stitch"; do run g4_s2_skip_${k//,/_} --insert-slices 2 --skip-kernels $k --no-kernel-timing; done
C="--no-clips --no-legs --no-cpu-baseline --steps 40 --gops-per-pipeline 1"
run g2_s4 --gops-per-call 2 --insert-slices 4
run g2_s2 --gops-per-call 2 --insert-slices 2
run g3_s2 --gops-per-call 3 --insert-slices 2
run g3_s3 --gops-per-call 3 --insert-slices 3
run g1_s4 --insert-slices 4
run g1_s16 --insert-slices 16
