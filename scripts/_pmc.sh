ROOT=$PWD
export ILQG_HIP_LIB=$ROOT/ilqgames_amd/libilqg_hip_a.so
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d /tmp/pmc1 -o x -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-latency > /tmp/pmc1.log 2>&1
tail -2 /tmp/pmc1.log | cut -c1-200
find /tmp/pmc1 -name "*.csv" | head
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmc1/**/*counter_collection.csv', recursive=True)
print(f)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k, {c: (len(x), sum(x)/len(x)) for c, x in v.items()})
PY
