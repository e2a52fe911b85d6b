mkdir -p gpurun_out/r3b; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "8192 f64 1" "8192 f64 0" "8192 f32 1" "8192 f32 0" "1024 f32 0"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3b/p_$1_$2_$3 -o p -- python $R/scripts/exp_modes.py --batch $1 --dtype $2 --split $3 --iters 5 --reps 2 > $R/gpurun_out/r3b/log_$1_$2_$3.txt 2>&1
  tail -1 $R/gpurun_out/r3b/log_$1_$2_$3.txt
  f=$(find $R/gpurun_out/r3b/p_$1_$2_$3 -name "*kernel_stats.csv" | head -1)
  head -8 $f | cut -c1-200
done
