( time timeout 600 python bench.py ) > gpurun_out/bench_default.txt 2>&1
timeout 300 python bench.py --baseline-config 3 --no-cpu-baseline --no-latency > gpurun_out/bench_c3.txt 2>&1
timeout 400 python bench.py --baseline-config 4 --no-cpu-baseline --no-latency > gpurun_out/bench_c4.txt 2>&1
timeout 400 python bench.py --baseline-config 4 --linesearch headline --no-cpu-baseline --no-latency > gpurun_out/bench_c4h.txt 2>&1
timeout 400 python bench.py --baseline-config 5 --no-cpu-baseline --no-latency > gpurun_out/bench_c5.txt 2>&1
tail -c 3000 gpurun_out/bench_default.txt; for f in c3 c4 c4h c5; do echo; echo == $f; tail -c 1500 gpurun_out/bench_$f.txt; done
