#!/bin/bash
# The rocprofv3 evidence of a round, one pass per BASELINE config bench.py reports (run on the GPU box via gpurun):
#   gpurun -- 'bash scripts/profile_all.sh r02'
# leaves gpurun_out/profiles/<tag>_*.md; copy the ones to be judged into profiles/.
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/profiles
run() {  # name, bench args
  export BENCH_ARGS="$2"
  rm -rf $ROOT/gpurun_out/prof
  bash $ROOT/scripts/profile.sh > /dev/null 2>&1
  (cd $ROOT && python scripts/summarize_profile.py gpurun_out/prof $TAG > gpurun_out/profiles/${TAG}_$1.md 2>&1)
}
run headline_f64_b1024 "--steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-latency --no-second-workload --no-copy-bandwidth"
run headline_f32_b1024 "--steps 10 --warmup 2 --repeats 1 --dtype f32 --no-cpu-baseline --no-latency --no-second-workload --no-copy-bandwidth"
run headline_f64_b8192 "--steps 10 --warmup 2 --repeats 1 --batch 8192 --no-cpu-baseline --no-latency --no-second-workload --no-copy-bandwidth"
run config3_f32_b8192 "--baseline-config 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-latency --no-second-workload --no-copy-bandwidth"
run config4_roundabout_T150_f64_b4096 "--baseline-config 4 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency --no-second-workload --no-copy-bandwidth"
run config5_reachability_f64_b2048 "--baseline-config 5 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency --no-second-workload --no-copy-bandwidth"
# beside the BASELINE configurations: the n = 16 constrained form of config 2 (the example BASELINE.json names) and the
# n = 24 game on the feedback sweep
run config2_n16_constrained_f64_b1024 "--config three_player_intersection --steps 6 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency --no-second-workload --no-copy-bandwidth"
run roundabout_feedback_n24_f64_b1024 "--config roundabout_merging_feedback --steps 6 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency --no-second-workload --no-copy-bandwidth"
rm -rf $ROOT/gpurun_out/prof
ls -la $ROOT/gpurun_out/profiles
