timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_nash.py -q -x 2>&1 | tail -4
timeout 600 python scripts/stage_bench.py --config three_player_overtaking 2>&1 | grep "lq_feedback"
timeout 600 python bench.py --config three_player_overtaking --steps 5 --warmup 1 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 | cut -c1-200
timeout 600 python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from ilqgames_amd import abi, examples, hip
for dt in (abi.F64, abi.F32):
    spec = examples.roundabout_merging(open_loop=False)
    spec.params.initial_alpha_scaling = 0.1; spec.params.expected_decrease_fraction = 0.001
    B = 1024
    prob = hip.Problem(spec, dt)
    x0 = torch.as_tensor(examples.jittered_x0(spec, B, seed=0), dtype=hip.torch_dtype(dt), device="cuda")
    bufs = prob.alloc_solve_buffers(B)
    prob.solve(x0, bufs, fixed_iters=2); torch.cuda.synchronize()
    for k in ("xs","us","P","alpha"): bufs[k].zero_()
    t0=time.perf_counter(); prob.solve(x0, bufs, fixed_iters=4); torch.cuda.synchronize(); t1=time.perf_counter()
    print("roundabout closed loop n=24 B=1024 T=100 dtype %d: %.2f ms per iteration of the batch" % (dt, (t1-t0)/4*1e3))
PY
