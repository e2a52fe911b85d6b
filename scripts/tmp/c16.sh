cd $GRAFT_REPO_ROOT; O=gpurun_out/c16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_generic.py tests/test_gpu_affine.py -q > $O/t.log 2>&1; tail -n 4 $O/t.log
BA="--no-cpu-baseline --no-latency --no-second-workload --repeats 3"
for a in "--config mixed_dubins_car_scene" "--config three_unicycle_scene" "--config mixed_dubins_car_scene_open_loop"; do
  python bench.py $BA $a 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$a]', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), round(d['mean_backtracks'],2), d['success_fraction'])"
done
