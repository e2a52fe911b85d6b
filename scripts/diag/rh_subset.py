import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from ilqgames_amd import abi, examples, hip
spec = examples.CONFIGS["three_player_collision_avoidance_reachability"]()
spec.params.max_solver_iters = 50
prob = hip.Problem(spec, abi.F64)
x0 = examples.jittered_x0(spec, 2048, seed=1)
tick = 0.5 * spec.T * spec.dt / 208
masks = []
def rec(r, info):
    if r == 1: masks.append(info["active"].cpu().numpy().astype(bool))
def run(x, tag):
    for rep in range(3):
        stamps = []
        def on(r, info):
            torch.cuda.synchronize(); stamps.append(time.perf_counter()); rec(r, info)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = prob.receding_horizon_simulate(x, final_time=1e9, planner_runtime=tick, extra_time=tick, solve_time=tick,
                                             augmented_lagrangian=True, max_records=13, on_record=on)
        torch.cuda.synchronize()
        print(tag, rep, "first %.0f ms, replans %.1f ms each" % ((stamps[0]-t0)*1e3, (stamps[-1]-stamps[0])/12*1e3), "active", int(out["active"].sum().item()), flush=True)
run(x0, "B=2048")
sub = x0[masks[0]]
print("subset", sub.shape)
run(sub, "B=%d" % sub.shape[0])
