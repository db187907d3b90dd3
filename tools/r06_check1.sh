#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06a; mkdir -p $O
( time python -m pytest tests/test_gpu_dist_shared.py -q -x -k "config3_world8 or config3_world2" ) > $O/dist_shared.txt 2>&1; tail -3 $O/dist_shared.txt
python bench.py --clip-frames 300 --steps 10 --warmup 2 --proxy 8,1 > $O/proxy81.json 2> $O/proxy81.err; tail -c 1500 $O/proxy81.json; tail -5 $O/proxy81.err
bash tools/r06_shard_sweep.sh > $O/shard_sweep.txt 2>&1; cat $O/shard_sweep.txt
SKIP_DEFAULT=1 bash tools/r06_profile.sh > $O/profile.log 2>&1; tail -150 $O/profile.log
