cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/p4 -- python /root/repo/bench.py --config roundabout_merging_T150 --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/p4.log 2>&1
tail -1 /root/repo/gpurun_out/p4.log | cut -c1-300
cd /root/repo
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/p4/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    print(r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
