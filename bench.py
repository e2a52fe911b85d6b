#!/usr/bin/env python
"""bench.py — batched iLQ iterations/s of the MI355X-native iLQGames solver.

One "step" = one outer iLQ iteration (src/ilq_solver.cpp:123-166: linearise + LQ Nash sweep +
(1+b) rollouts + (1+b) cost quadraticisations + total costs) of EVERY instance of the batch.
Default workload = BASELINE.json configs[1]: three-player intersection, n=14
(ModifiedThreePlayerIntersectionExample — the n=14 game), N=3, T=100, fp64,
batch=1024 instances per GPU, inputs resident in HBM before the timed region.

Multi-GPU (weak scaling): one process per GPU under torch.distributed.run; every rank solves its
own `batch` instances (no collective in the data path) and the converged strategies are gathered
to rank 0 over RCCL/xGMI inside the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def algorithmic_bytes_per_iteration(n, m, N, T, pairs_m, elem, backtracks=0.0):
    """s*T*[2(W_lin+W_quad) + 2 W_strat + 2 W_op + n] (+ per extra back-track
    s*T*[W_quad + W_strat + 2 W_op])."""
    W_lin = n * n + n * m
    W_quad = N * (n * n + n) + sum(mj * mj + mj for mj in pairs_m)
    W_strat = m * n + m
    W_op = n + m
    base = elem * T * (2 * (W_lin + W_quad) + 2 * W_strat + 2 * W_op + n)
    extra = elem * T * (W_quad + W_strat + 2 * W_op)
    return base + backtracks * extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--config", default="modified_three_player_intersection")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the single-instance ms/solve figure (profiling runs: keeps the kernel statistics to the batch)")
    ap.add_argument("--cpu-sample", type=int, default=768,
                    help="instances in the CPU baseline sample (768 x 20 iterations ~ 15 s on one host thread)")
    args = ap.parse_args()

    import torch
    from ilqgames_amd import abi, examples, hip

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dtype = abi.F64 if args.dtype == "f64" else abi.F32
    elem = 8 if dtype == abi.F64 else 4

    # Workload.  The n=14 example's own line-search fraction (0.9) makes the REFERENCE's line search
    # fail at iteration 2 (tests/test_gpu_parity.py::test_ilq_solve_free_running...), so throughput
    # is measured with the line-search parameters of exec/three_player_intersection/main.cpp:109-120
    # (alpha0 = 0.1, fraction 0.001) at a fixed iteration count, as SURVEY.md §8(d) prescribes.
    spec = examples.CONFIGS[args.config]()
    spec.params.initial_alpha_scaling = 0.1
    spec.params.expected_decrease_fraction = 0.001
    spec.params.max_backtracking_steps = 100
    B = args.batch
    x0 = examples.jittered_x0(spec, B, seed=1000003 * rank)
    prob = hip.Problem(spec, dtype)
    x0_d = torch.as_tensor(x0, dtype=hip.torch_dtype(dtype), device="cuda")
    bufs = prob.alloc_solve_buffers(B)

    def reset():
        for k in ("xs", "us", "P", "alpha"):
            bufs[k].zero_()

    def solve(iters):
        prob.solve(x0_d, bufs, fixed_iters=iters)

    gathered = None
    if distributed:
        strat = torch.cat([bufs["P"].reshape(B, -1), bufs["alpha"].reshape(B, -1)], dim=1)
        gathered = [torch.empty_like(strat) for _ in range(world)] if rank == 0 else None

    def gather():
        if distributed:
            s = torch.cat([bufs["P"].reshape(B, -1), bufs["alpha"].reshape(B, -1)], dim=1)
            dist.gather(s, gathered, dst=0)

    # warmup (untimed): W iterations
    reset()
    solve(max(1, args.warmup))
    gather()
    torch.cuda.synchronize()

    # timed: exactly K iterations of every instance = K (LQ kernel, trial kernel) rounds after the
    # initial trial pass; all launches are enqueued back to back on the current stream
    reset()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1, ev2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0 = time.perf_counter()
    ev0.record()
    solve(args.steps)
    ev1.record()
    gather()
    ev2.record()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_s = ev0.elapsed_time(ev1) * 1e-3
    gather_s = ev1.elapsed_time(ev2) * 1e-3
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    iters = bufs["iters"].cpu().numpy()
    status = bufs["status"].cpu().numpy()
    total_iters = int(iters.sum())
    if distributed:
        ti = torch.tensor([total_iters], dtype=torch.int64, device="cuda")
        dist.all_reduce(ti)
        total_iters = int(ti.item())

    if rank == 0:
        n, m, N, T = prob.n, prob.m, prob.N, prob.T
        pairs_m = [spec.udims[j] for _, j in prob.pairs]
        bytes_iter = algorithmic_bytes_per_iteration(n, m, N, T, pairs_m, elem)
        value = total_iters / elapsed
        # Roofline of the hot path on this rank.  One outer iteration of the batch is one round of two
        # kernels (ilq_lq_kernel: Riccati sweep; ilq_trial_kernel: rollout + linearise/quadraticise +
        # line-search decision); the pair is the "launch" the algorithmic bytes are counted for, timed
        # with HIP events over the K rounds (profiles/: the two kernels' rocprofv3 averages add up to it).
        launch_bytes = bytes_iter * int(iters.sum())
        achieved = launch_bytes / kernel_s / 1e9
        # HBM traffic per round from the committed PMC passes of this workload (bench.py cannot run rocprofv3
        # on itself; scripts/profile.sh collects the counters, profiles/traffic.json holds their medians)
        traffic = None
        try:
            tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")))
            ent = tj.get("%s:%s:%d" % (args.config, args.dtype, B))
            if ent:
                traffic = ent["bytes_per_round"]
        except (OSError, ValueError, KeyError):
            traffic = None
        out = {
            "metric": "iLQ iterations/sec (batch)", "value": value, "unit": "instance-iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s n=%d N=%d T=%d batch=%d/GPU %s, fixed %d outer iterations, "
                                   "alpha0=0.1 frac=0.001" % (args.config, n, N, T, B, args.dtype, args.steps),
                       "parallelism": "instances sharded across %d GPU(s); RCCL gather of strategies" % world},
            "ms_per_solve_batch": kernel_s * 1e3,
            "gather_ms": gather_s * 1e3,
            "success_fraction": float(status.mean()),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "ilq_lq_kernel + ilq_trial_kernel (one round = one outer iteration of the batch)",
                         "launch_ms": kernel_s * 1e3 / max(1, args.steps),
                         "algorithmic_bytes_per_launch": launch_bytes / max(1, args.steps),
                         "bytes_per_iteration_per_instance": bytes_iter},
        }
        if world == 1 and not args.no_latency:
            # BASELINE.json's second figure, ms per solve: ONE instance run to its convergence test (free-running:
            # the host reads the instance's state back after every kernel round), zero warm start, outside the
            # timed region above.
            lb = prob.alloc_solve_buffers(1)
            prob.solve(x0_d[:1], lb)
            lat = []
            for _ in range(3):
                for k in ("xs", "us", "P", "alpha"):
                    lb[k].zero_()
                torch.cuda.synchronize()
                l0 = time.perf_counter()
                prob.solve(x0_d[:1], lb)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - l0)
            out["latency"] = {"ms_per_solve": sorted(lat)[1] * 1e3, "iterations": int(lb["iters"][0].item()),
                              "converged": int(lb["converged"][0].item()), "success": int(lb["status"][0].item()),
                              "instances": 1, "mode": "free-running to convergence_tolerance, zero warm start"}
        if not args.no_cpu_baseline and world == 1:
            from oracle import pyoracle
            S = min(args.cpu_sample, B)
            op = pyoracle.OracleProblem(spec)
            op.solve(dtype, x0[:2], fixed_iters=2)  # warm the code path
            c0 = time.perf_counter()
            ref = op.solve(dtype, x0[:S], fixed_iters=args.steps, threads=1)
            c1 = time.perf_counter()
            cpu_iters = int(ref["iters"].sum())
            out["cpu_baseline"] = {
                "value": cpu_iters / (c1 - c0), "unit": "instance-iterations/s", "cores": 1, "kind": "port",
                "sample": "%d of the same %d instances x %d iterations, oracle/ilqg_oracle.hpp (%s), 1 thread — "
                          "the reference's execution model; the reference binary itself cannot be built here "
                          "(Eigen3/glog/gflags absent)" % (S, B, args.steps, args.dtype),
                "seconds": c1 - c0,
            }
            ncpu = os.cpu_count() or 1
            if ncpu > 1:
                c0 = time.perf_counter()
                ref2 = op.solve(dtype, x0[:min(B, S * 4)], fixed_iters=args.steps, threads=ncpu)
                c1 = time.perf_counter()
                out["cpu_baseline"]["value_all_cores"] = int(ref2["iters"].sum()) / (c1 - c0)
                out["cpu_baseline"]["cores_all"] = ncpu
            c0 = time.perf_counter()
            one = op.solve(dtype, x0[:1])
            if "latency" in out:
                out["latency"]["cpu_ms_per_solve"] = (time.perf_counter() - c0) * 1e3
                out["latency"]["cpu_iterations"] = int(one["iters"][0])
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
