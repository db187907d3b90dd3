# samples the shader clock while the query benchmark loops (profiling aid)
cd $GRAFT_REPO_ROOT
(for i in 1 2 3 4 5 6; do ./build/bench_query > /dev/null 2>&1; done) &
BG=$!
for i in 1 2 3 4 5 6 7 8; do
  sleep 0.7
  rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -4
  rocm-smi --showpower 2>/dev/null | grep -i "power" | head -2
  echo ---
done
wait $BG
rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2
