#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_nt4; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
run() { name=$1; shift; timeout 300 python bench.py --no-clips --no-legs --no-cpu-baseline --steps 20 --warmup 5 "$@" > $O/$name.json 2> $O/$name.err; }
run g1
run g4 --gops-per-call 4 --gops-per-pipeline 1
run g1b
run g4b --gops-per-call 4 --gops-per-pipeline 1
run g1_u16 --bits 16
python3 tools/show_bench.py $O/*.json
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_nt4/bench_default.json').read().splitlines() if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['regions_ms'], d['steady_state']['value'])
for k in ('decode_1080p','interleaved_yuv444','batched_gops','config4_2160p','config4_2160p_gop9','clip300','clip300_uint16'):
    v=d.get(k); print(k, v.get('value'), v.get('ms_per_step') or v.get('ms_per_pass') or v.get('ms_per_gop'))
PY
