cd $GRAFT_REPO_ROOT; O=gpurun_out/c9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_single_wave.py -q -x > $O/t1w.log 2>&1; tail -n 4 $O/t1w.log
BA="--no-cpu-baseline --no-latency --no-second-workload --repeats 3"
for a in "--batch 8192 --steps 10" "--baseline-config 3" "--batch 2048" "--dtype f32 --batch 2048"; do
  python bench.py $BA $a 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"
done
