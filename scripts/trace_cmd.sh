#!/bin/bash
# Top kernels (calls, total us, mean us, share) of any command under rocprofv3 --kernel-trace --stats, on the GPU box:
#   bash scripts/trace_cmd.sh TAG python scripts/mpc_bench.py --al --steps 12
ROOT=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is unset)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( cd "$ROOT" && rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o run -- "$@" > "$OUT/log" 2>&1 )
cd "$ROOT" || exit 1
tail -n 2 "$OUT/log"
python - "$OUT/trace/run_results.db" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print("kernel launches: %d, kernel time %.1f ms" % (sum(r[1] for r in rows), sum(r[2] for r in rows) / 1e6))
for r in rows[:14]:
    print("%-74s %6d %10.1f %9.1f %6.2f" % (r[0][:74].replace("void (anonymous namespace)::", ""), r[1], r[2] / 1e3, r[3] / 1e3, r[4]))
PY
