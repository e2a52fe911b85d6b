#!/bin/bash
# A/B of the round counters' read-back (ILQG_READBACK=copy: hipMemcpyAsync + hipStreamSynchronize; default: a publishing
# kernel + a host spin, read_round_counters in csrc/ilqg_api.hip) on the host-counted workloads, on the GPU box:
#   gpurun -- bash scripts/ab_readback.sh
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is unset)}" || exit 1
O=gpurun_out/ab_readback; mkdir -p $O; : > $O/ab.log
BA="--no-cpu-baseline --no-latency --no-second-workload --no-configs --no-copy-bandwidth --repeats 5"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['mean_backtracks'])"; }
for rep in 1 2; do for mode in copy spin; do
  if [ $mode = copy ]; then export ILQG_READBACK=copy; else unset ILQG_READBACK; fi
  timeout 200 python bench.py $BA --config three_player_intersection --steps 6 2>>$O/err.log | tail -n 1 | line n16 $mode >> $O/ab.log
  timeout 200 python bench.py $BA --baseline-config 5 2>>$O/err.log | tail -n 1 | line c5scene $mode >> $O/ab.log
  timeout 200 python bench.py $BA --baseline-config 4 2>>$O/err.log | tail -n 1 | line c4 $mode >> $O/ab.log
  timeout 300 python scripts/mpc_bench.py --al --steps 12 2>>$O/err.log | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rh_al', '$mode', round(d['ms_per_call'],1), d['logged_iterates'], d['active_per_call'])" >> $O/ab.log
done; done
cat $O/ab.log
