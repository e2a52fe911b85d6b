cd $GRAFT_REPO_ROOT; O=gpurun_out/c22; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_single_wave.py tests/test_gpu_parity.py tests/test_gpu_forced.py tests/test_gpu_fullsize.py -q > $O/t.log 2>&1; tail -n 3 $O/t.log
BA="--no-cpu-baseline --no-latency --no-second-workload --repeats 3"
for a in "" "--batch 8192 --steps 10" "--baseline-config 3" "--dtype f32"; do
  python bench.py $BA $a 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$a]', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"
done
