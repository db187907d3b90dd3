#!/bin/bash
# Round 6: what ONE rank of an N-way split of the 300-frame clip does on one GPU (bench.py --clip-frames 300 --proxy N): every rank of the
# split in turn, gather stubbed.  slots1 = one set of coders (round 5: a pass's stages one after the other on one pipeline),
# auto = pass slots (consecutive passes rotate over pipelines // blocks sets), gK = the share cut into K equal blocks per pass.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
run() {  # label, args...
  local label=$1; shift
  python bench.py --clip-frames 300 --steps 40 --warmup 2 --no-verify "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['slowest']
print('%-34s slowest rank %d: %.4f ms/pass sustained, %.4f ms one pass alone, blocks %s, pass slots %d | per rank %s' % ('$label', s['proxy_of']['rank'], s['ms_per_pass'], s['pass_latency_ms'], s['blocks_rank0'], s['pass_slots'], d['ms_per_pass_per_rank']))"
}
for bits in 8 16; do
  for N in 8 4 2; do
    run "u$bits N=$N slots1"        --bits $bits --proxy $N --clip-pass-slots 1
    run "u$bits N=$N auto"          --bits $bits --proxy $N
    run "u$bits N=$N slots2"        --bits $bits --proxy $N --clip-pass-slots 2
  done
  run "u$bits N=8 g2 slots1"        --bits $bits --proxy 8 --clip-groups 2 --clip-pass-slots 1
  run "u$bits N=8 g4 slots1"        --bits $bits --proxy 8 --clip-groups 4 --clip-pass-slots 1
  run "u$bits N=8 g2 auto"          --bits $bits --proxy 8 --clip-groups 2
  run "u$bits N=4 g2 auto"          --bits $bits --proxy 4 --clip-groups 2
  run "u$bits N=4 g4 slots1"        --bits $bits --proxy 4 --clip-groups 4 --clip-pass-slots 1
  python bench.py --clip-frames 300 --steps 40 --warmup 2 --no-verify --bits $bits 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('u$bits N=1: %.4f ms/pass (%.0f Mpixel/s)' % (d['ms_per_step'], d['value']))"
done
