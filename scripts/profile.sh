#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's roofline block (run on the GPU box via gpurun).
#   1. kernel trace + stats           -> gpurun_out/prof/trace
#   2. PMC pass: FETCH_SIZE           -> gpurun_out/prof/pmc_fetch   (own run, no trace domains)
#   3. PMC pass: WRITE_SIZE           -> gpurun_out/prof/pmc_write
# Summaries are copied into profiles/ by hand afterwards (profiles/ is tracked, gpurun_out/ is scratch).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
ARGS=${BENCH_ARGS:---steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-latency --no-second-workload --no-copy-bandwidth}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $ROOT/bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- python $ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- python $ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1
#   4. PMC pass: wave-level SQ counters (issue / wait / busy)  -> gpurun_out/prof/pmc_sq
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq -o bench -- python $ROOT/bench.py $ARGS > $OUT/bench_sq.log 2>&1
cd $ROOT
python bench.py ${ARGS/--repeats 1/--repeats 5} > $OUT/bench_plain.log 2>&1
find $OUT -type f | head -50
tail -2 $OUT/bench_plain.log
