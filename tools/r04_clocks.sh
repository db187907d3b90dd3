cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
( python bench.py --no-cpu-baseline --no-clips --no-legs --no-kernel-timing --no-verify --steps 150000 --exact-steps > $O/bench_long.json 2>/dev/null ) &
BP=$!
sleep 5
for i in $(seq 1 40); do
  echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' ')"
  sleep 0.4
done > $O/clocks_under_load.txt
wait $BP
cat $O/clocks_under_load.txt | cut -c1-160
python -c "import json;d=json.loads(open('$O/bench_long.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['steps'])"
