ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pick_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --kernel-include-regex "ilq_probe_merit" -d $OUT/a -o pick --output-format csv -- python $ROOT/scripts/mpc_bench.py --al --steps 3 > $OUT/log_a 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --kernel-include-regex "ilq_probe_merit" -d $OUT/b -o pick --output-format csv -- python $ROOT/scripts/mpc_bench.py --al --steps 3 > $OUT/log_b 2>&1
cd $ROOT
find $OUT -name "*.csv" | head
python - <<'PY'
import csv,glob,collections,os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pick_pmc'
for f in sorted(glob.glob(root+'/*/*counter_collection.csv')+glob.glob(root+'/*/*/*counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    print(f, len(rows), rows[0].keys() if rows else None)
    agg=collections.defaultdict(list)
    for r in rows:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        v.sort(); print("  %-24s n %4d median %12.0f  last %12.0f"%(k,len(v),v[len(v)//2],v[-1]))
PY
