cd $GRAFT_REPO_ROOT; O=gpurun_out/c15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_single_wave.py -q -k "split or schedule or probe or handoff" > $O/t.log 2>&1; tail -n 3 $O/t.log
BA="--no-cpu-baseline --no-latency --no-second-workload --repeats 3"
for a in "--config three_player_intersection --steps 6" "--baseline-config 5" "--baseline-config 4"; do
  python bench.py $BA $a 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$a]', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), round(d['mean_backtracks'],2))"
done
