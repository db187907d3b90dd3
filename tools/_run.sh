mkdir -p gpurun_out/r03final
( echo '$ SOAK_BIG=0.6 python tools/fuzz_soak.py 200 20260927'; SOAK_BIG=0.6 timeout 400 python tools/fuzz_soak.py 200 20260927; echo '$ SOAK_BIG=1.0 python tools/fuzz_soak.py 100 3'; SOAK_BIG=1.0 timeout 300 python tools/fuzz_soak.py 100 3; echo '$ python tools/fuzz_surface.py 60'; timeout 200 python tools/fuzz_surface.py 60 ) > gpurun_out/r03final/fuzz_soak.txt 2>&1
cat gpurun_out/r03final/fuzz_soak.txt | tail -12
