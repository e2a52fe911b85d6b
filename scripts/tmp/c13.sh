cd $GRAFT_REPO_ROOT; O=gpurun_out/c13; mkdir -p $O
export ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_ol.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forced.py -q -k "open or roundabout or config4 or ol_" > $O/t.log 2>&1; tail -n 5 $O/t.log
BA="--no-cpu-baseline --no-latency --no-second-workload --repeats 3"
for a in "--baseline-config 4"; do
  python bench.py $BA $a 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$a]', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), round(d['mean_backtracks'],2))"
done
unset ILQG_HIP_LIB
python bench.py $BA --baseline-config 4 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[base config 4]', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), round(d['mean_backtracks'],2))"
