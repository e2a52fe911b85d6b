"""Receding-horizon harness throughput (BASELINE config 5 shape): B instances x K warm-started solver calls,
everything resident on the device.  Prints one JSON line.  Usage:
  python scripts/mpc_bench.py [--config NAME] [--batch 2048] [--steps 200] [--al]
The simulator's clock advances `tick` seconds of extra time plus `tick` seconds of simulated solve time per call
(the reference uses 0.25 + wall clock; 200 calls inside a 10 s horizon need 0.025)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="three_player_collision_avoidance_reachability")
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--al", action="store_true")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--max-iters", type=int, default=50,
                    help="SolverParams::max_solver_iters of every call (a call costs ~1.2 ms per outer iteration "
                         "whatever the batch size, so the reference's 1000 makes a 200-call run take minutes)")
    a = ap.parse_args()
    import torch
    from ilqgames_amd import abi, examples, hip
    spec = examples.CONFIGS[a.config]()
    spec.params.max_solver_iters = a.max_iters
    dtype = abi.F64 if a.dtype == "f64" else abi.F32
    prob = hip.Problem(spec, dtype)
    x0 = examples.jittered_x0(spec, a.batch, seed=1)
    tick = 0.5 * spec.T * spec.dt / (a.steps + 8)
    iters, active = [], []

    def on_record(r, info):
        iters.append(info["bufs"]["iters"].sum().item())
        active.append(int(info["active"].sum().item()))

    torch.cuda.synchronize()
    t0 = time.time()
    out = prob.receding_horizon_simulate(x0, final_time=1e9, planner_runtime=tick, extra_time=tick, solve_time=tick,
                                         augmented_lagrangian=a.al, max_records=a.steps + 1, on_record=on_record)
    torch.cuda.synchronize()
    wall = time.time() - t0
    solves = int(out["num_records"].sum().item())
    print(json.dumps(dict(config=a.config, dtype=a.dtype, batch=a.batch, solver="al" if a.al else "ilq",
                          max_solver_iters=a.max_iters, calls=out["calls"], instance_solves=solves, seconds=wall,
                          ms_per_call=1e3 * wall / out["calls"], instance_solves_per_s=solves / wall,
                          logged_iterates=int(sum(iters)), active_at_end=int(out["active"].sum().item()),
                          active_per_call=active[:3] + active[-2:],
                          spliced_plans=int((out["plan"]["len"] > spec.T).sum().item()))))


if __name__ == "__main__":
    main()
