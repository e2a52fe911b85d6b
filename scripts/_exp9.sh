cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
ILQG_HIP_LIB=$R/ilqgames_amd/libilqg_hip_ol.so timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p4 -o p -- python $R/scripts/exp_modes.py --config roundabout_merging_T150 --batch 4096 --dtype f64 --iters 4 --reps 2 > $R/gpurun_out/p4.log 2>&1
cd $R; python - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/p4/**/*.db', recursive=True)[0]
con = sqlite3.connect(db)
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()[:10]:
    print("%-60s %4d %10.1f %9.1f %6.2f" % (r[0][:60], r[1], r[2], r[3], r[4]))
PY
tail -1 gpurun_out/p4.log
