#!/bin/bash
# Whole-path throughput beyond 1080p: which kernel family (LDS-resident, LDS-tiled, global-memory) the plan
# picks and what it costs.  Usage (GPU box): tools/large_frames.sh > gpurun_out/large_frames.txt
run() {
    timeout 300 python bench.py --no-cpu-baseline --no-clips --no-legs --steps 10 --warmup 2 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
s = d.get('steady_state') or d
print('%-62s %8.0f Mpixel/s %8.3f ms/step  %s  verified %s' % ('$*', s['value'], s['ms_per_step'], ' '.join('%s=%.0f' % (a, b * 1e3) for a, b in d['kernels_ms_per_step_alone'].items()), d['verified_vs_oracle']['frames']))"
}
run --width 1920 --height 1080 --frames 30
run --width 1920 --height 1080 --frames 30 --gops-per-call 4 --gops-per-pipeline 1
run --width 2560 --height 1440 --frames 30
run --width 3840 --height 2160 --frames 30 --gops-per-pipeline 1
run --width 3840 --height 2160 --frames 9
run --width 5120 --height 2880 --frames 9 --gops-per-pipeline 1
run --width 7680 --height 4320 --frames 5 --gops-per-pipeline 1
