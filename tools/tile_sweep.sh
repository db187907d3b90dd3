#!/bin/bash
# BASELINE config 4: 3840x2160 YUV444 synthetic residuals, LDS tile-size sweep (runs on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
echo "tile_KiB  Mpixel/s  ms/step  insert_ms  query_ms   (3840x2160, 8 inter-frames per step, 1 pipeline)"
for t in 4 8 16 32 64 96 128 0; do
  python $ROOT/bench.py --width 3840 --height 2160 --frames 9 --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --lds-tile-kib $t 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('%-9s %-9.0f %-8.3f %-10.3f %-8.3f' % ('$t' if '$t'!='0' else 'auto', d['value'], d['ms_per_step'], k['insert'], k['query']))"
done
