#!/bin/bash
# round 5: the clip legs (BASELINE configs[2] / [4] at N = 1) under different block partitions; then the rocprofv3 evidence run
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_clip; mkdir -p $O
run() { name=$1; shift; timeout 300 python bench.py --clip-frames 300 --steps 10 --warmup 2 "$@" > $O/$name.json 2> $O/$name.err; python3 -c "
import json,sys
d=json.loads([l for l in open('$O/$name.json').read().splitlines() if l.startswith('{')][-1])
print('%-28s %9.0f Mpx/s  %.4f ms/pass  verified %s' % ('$name', d['value'], d['ms_per_step'], d['verified_vs_oracle']['frames']))" ; }
run auto_p4
run auto_p4_u16 --bits 16
run auto_p3 --streams 3
run auto_p5 --streams 5
run bg2_p5 --clip-block-gops 2 --streams 5
run bg2_p4 --clip-block-gops 2 --streams 4
run bg1_p4 --clip-block-gops 1
run bg5_p2 --clip-block-gops 5 --streams 2
bash tools/r05_profile.sh > $O/profile.log 2>&1; tail -60 $O/profile.log
