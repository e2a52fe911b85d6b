cd $GRAFT_REPO_ROOT; ROOT=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
export BENCH_ARGS="--config three_player_intersection --steps 6 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency --no-second-workload"
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof/trace -o bench -- python $ROOT/bench.py $BENCH_ARGS > $ROOT/gpurun_out/prof/bench_trace.log 2>&1
cd $ROOT
python - <<'PY'
import sqlite3, os
con = sqlite3.connect('gpurun_out/prof/trace/bench_results.db')
cur = con.cursor()
for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()[:14]:
    print("%-90s %5d %10.1f %9.1f %6.2f" % (r[0][:90], r[1], r[2], r[3], r[4]))
rows = cur.execute("select name, duration, grid_x from kernels where name like '%ilq_%' order by start").fetchall() if False else []
PY
