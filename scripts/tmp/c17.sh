cd $GRAFT_REPO_ROOT
for rep in 1 2; do for tag in base gj; do
  if [ $tag = base ]; then unset ILQG_HIP_LIB; else export ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_$tag.so; fi
  python bench.py --no-cpu-baseline --no-second-workload --no-configs --repeats 5 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), 'latency', round(d['latency']['ms_per_iteration'],4), d['latency']['iterations'])"
done; done
export ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_gj.so
python scripts/quick_parity.py 2>&1 | grep -v amdgpu | tail -n 9
