ROOT=$PWD
export ILQG_HIP_LIB=$ROOT/ilqgames_amd/libilqg_hip_a.so
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_IFETCH"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pm$i -o x -- python $ROOT/scripts/_q.py > /tmp/pm$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pm*/x_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'rows_kernel' in k and 'ilq_' not in k:
            agg[(k[:50], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   %-24s n=%d median=%.4g" % (c, len(x), sorted(x)[len(x)//2]))
PY
