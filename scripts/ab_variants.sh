#!/bin/bash
# A/B of experiment builds (scripts/devbuild.py --tag T ...) on the GPU box: gpurun -- "PARITY_TAGS=\"t1 t2\" bash scripts/ab_variants.sh OUTDIR base t1 t2"
# runs bench.py twice per tag (tag "base" = the product library), then scripts/quick_parity.py on the PARITY_TAGS.
# usage: ab.sh OUTDIR tag1 tag2 ...   (tag "base" = product library); runs each twice, then quick parity on the last tag
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is unset)}" || exit 1
O=gpurun_out/$1; shift; mkdir -p $O
BA="${BENCH_ARGS:---no-cpu-baseline --no-latency --no-second-workload --repeats 5}"
for rep in 1 2; do for tag in "$@"; do
  if [ $tag = base ]; then unset ILQG_HIP_LIB; else export ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_$tag.so; fi
  timeout 120 python bench.py $BA 2>$O/err_$tag.log | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['mean_backtracks'])" >> $O/ab.log 2>&1
done; done
for tag in ${PARITY_TAGS:-${@: -1}}; do
export ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_$tag.so
echo "== parity $tag" >> $O/parity.log
timeout 200 python scripts/quick_parity.py >> $O/parity.log 2>&1
done
cat $O/ab.log; grep -v "amdgpu.ids" $O/parity.log | tail -n 12
