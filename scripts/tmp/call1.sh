#!/bin/bash
# GPU call 1: new tests, baseline + A/B of sweep experiments, phase profile, timeline
cd $GRAFT_REPO_ROOT
O=gpurun_out/c1; mkdir -p $O
BA="--no-cpu-baseline --no-latency --no-second-workload --repeats 5"
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "max_runtime or equality_flag" > $O/tests.log 2>&1
for tag in base sw cc all base sw cc all; do
  if [ $tag = base ]; then unset ILQG_HIP_LIB; else export ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_$tag.so; fi
  timeout 120 python bench.py $BA 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']), d['ms_per_step'], d['roofline']['frac'])" >> $O/ab.log 2>&1
done
export ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_all.so
timeout 200 python scripts/quick_parity.py > $O/parity_all.log 2>&1
export ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_prof.so
timeout 200 python scripts/stage_bench.py > $O/stage_prof.log 2>&1
export ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_tl.so
timeout 200 python scripts/timeline.py > $O/timeline.log 2>&1
unset ILQG_HIP_LIB
cat $O/tests.log | tail -n 3; cat $O/ab.log; tail -n 5 $O/parity_all.log
