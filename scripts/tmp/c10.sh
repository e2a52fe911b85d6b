cd $GRAFT_REPO_ROOT
BA="--no-cpu-baseline --no-latency --no-second-workload --repeats 3"
for a in "" "--single-wave on" "--single-wave on --split-trial off" "--single-wave on --split-trial off --adjoint off" "--single-wave on --split-trial off --adjoint on" "--batch 1536" "--batch 1536 --single-wave off" "--batch 1280" "--batch 1280 --single-wave off"; do
  python bench.py $BA $a 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$a]', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"
done
