#!/bin/bash
# First GPU call of round 2: baseline numbers + phase profile + trig micro-benchmark (diagnostic).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
{
echo "== ubench"; timeout 60 scripts/ubench/_bin/trig_lat; timeout 60 scripts/ubench/_bin/mfma_lat
echo "== bench baseline f64 b1024"; timeout 300 python bench.py --no-cpu-baseline --no-latency
for b in 2048 8192; do echo "== bench f64 b$b"; timeout 300 python bench.py --no-cpu-baseline --no-latency --batch $b --steps 10; done
echo "== bench f32 b1024"; timeout 300 python bench.py --no-cpu-baseline --no-latency --dtype f32
echo "== stage bench (profile build)"; ILQG_HIP_LIB=$ROOT/ilqgames_amd/libilqg_hip_prof.so timeout 300 python scripts/stage_bench.py
} > $OUT/diag1.log 2>&1
tail -40 $OUT/diag1.log
