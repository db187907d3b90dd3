#!/bin/bash
# Whole-path throughput beyond 1080p: which kernel family (LDS-resident, LDS-tiled, global-memory) the plan
# picks and what it costs.  Usage (GPU box): tools/large_frames.sh > gpurun_out/large_frames.txt
run() {
    timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" 2>&1 | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('%-70s %8.0f Mpixel/s %8.3f ms/step  %s' % ('$*', d['value'], d['ms_per_step'], ' '.join('%s=%.3f' % kv for kv in d['kernels_ms_per_step'].items())))"
}
run --width 1920 --height 1080 --frames 30
run --width 2560 --height 1440 --frames 17
run --width 3840 --height 2160 --frames 9
run --width 3840 --height 2160 --frames 9 --generic-kernels
run --width 5120 --height 2880 --frames 7
run --width 5120 --height 2880 --frames 7 --lds-tile-kib 160
run --width 5120 --height 2880 --frames 7 --generic-kernels
run --width 6016 --height 3384 --frames 5
run --width 6016 --height 3384 --frames 5 --lds-tile-kib 160
run --width 7680 --height 4320 --frames 5
run --width 7680 --height 4320 --frames 5 --lds-tile-kib 160
run --width 7680 --height 4320 --frames 5 --generic-kernels
run --width 15360 --height 8640 --frames 3
run --width 15360 --height 8640 --frames 3 --lds-tile-kib 160
