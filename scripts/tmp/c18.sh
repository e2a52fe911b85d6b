cd $GRAFT_REPO_ROOT
for rep in 1 2; do for tag in base occ2; do
  if [ $tag = base ]; then unset ILQG_HIP_LIB; else export ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_$tag.so; fi
  python bench.py --dtype f32 --no-cpu-baseline --no-second-workload --no-latency --repeats 5 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"
done; done
