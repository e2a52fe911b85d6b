# Top kernels of one bench.py configuration under rocprofv3 --kernel-trace --stats (GPU box): bash scripts/trace_top.sh "--dtype f32"
# usage: trace.sh "<bench args>"  -> prints top kernels
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is unset)}
cd "$ROOT" || exit 1
rm -rf gpurun_out/prof_t; mkdir -p gpurun_out/prof_t
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_t/trace -o bench -- python $ROOT/bench.py $1 --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-latency --no-second-workload --no-copy-bandwidth > $ROOT/gpurun_out/prof_t/log 2>&1
cd $ROOT
python - <<'PY'
import sqlite3
con = sqlite3.connect('gpurun_out/prof_t/trace/bench_results.db')
cur = con.cursor()
for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()[:6]:
    print("%-80s %5d %10.1f %9.1f %6.2f" % (r[0][:80], r[1], r[2], r[3], r[4]))
PY
