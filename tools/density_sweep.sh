#!/bin/bash
# SURVEY 8d secondary table: whole-path throughput over the changed-pixel density (1080p, 29 inter-frames).
# Usage (on the GPU box): tools/density_sweep.sh > gpurun_out/density_sweep.txt
for p in 0.01 0.05 0.08889 0.2 0.3; do
    python bench.py --density $p --no-cpu-baseline --steps 30 --warmup 3 | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
k = d['kernels_ms_per_step']
print('p=%-8s %9.0f Mpixel/s  %.4f ms/step  kernels(ms, alone): %s' % ('$p', d['value'], d['ms_per_step'], ' '.join('%s=%.4f' % kv for kv in k.items())))"
done
