#!/bin/bash
# SURVEY 8d secondary table: whole-path throughput over the changed-pixel density (1080p, 29 inter-frames per step, four pipelines).
# Usage (on the GPU box): tools/density_sweep.sh > gpurun_out/density_sweep.txt
for p in 0.01 0.05 0.08889 0.2 0.3; do
    python bench.py --density $p --no-cpu-baseline --no-clips --no-legs --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
k = d['kernels_ms_per_step_alone']
print('p=%-8s %9.0f Mpixel/s (median 20-step region)  %9.0f (steady)  %.4f ms/step  kernels alone (us): %s  verified %s' % ('$p', d['value'], d['steady_state']['value'], d['steady_state']['ms_per_step'], ' '.join('%s=%.1f' % (a, b * 1e3) for a, b in k.items()), d['verified_vs_oracle']['frames']))"
done
