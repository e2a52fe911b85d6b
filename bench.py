#!/usr/bin/env python
"""bench.py — batched iLQ iterations/s of the MI355X-native iLQGames solver.

One "step" = one outer iLQ iteration (src/ilq_solver.cpp:123-166: linearise + LQ Nash sweep +
(1+b) rollouts + (1+b) cost quadraticisations + total costs) of EVERY instance of the batch.
Default workload = BASELINE.json configs[1]: three-player intersection, n=14
(ModifiedThreePlayerIntersectionExample — the n=14 game), N=3, T=100, fp64,
batch=1024 instances per GPU, inputs resident in HBM before the timed region.

Multi-GPU (weak scaling): one process per GPU under torch.distributed.run; every rank solves its
own `batch` instances (no collective in the data path) and the converged strategies are gathered
to rank 0 (ilqgames_amd/sharding.py: RCCL over xGMI) inside the timed region.

Prints ONE JSON line on rank 0.

`--backend stub` replaces the device solve by a CPU stand-in over gloo: it exists so that the N > 1 code of
this file (sharding, gather, max-over-ranks timing, the JSON line) is exercised by a 2-rank CPU test
(tests/test_host_boundary.py); it measures nothing.
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
# Dense fp64 rate, DERIVED (the guide lists no fp64 figure): half of its fp32 vector peak of 157.3 TFLOP/s, i.e.
# 32 flop/clk/SIMD — what scripts/ubench/mfma_lat.hip measures for v_mfma_f64_16x16x4_f64 (64 cycles per 2048 flop)
# and for dependent-free v_fma_f64 streams on this part.
FP32_PEAK_TFLOPS = 157.3
FP64_PEAK_TFLOPS = FP32_PEAK_TFLOPS / 2


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


def csrc_sha16():
    """Identity of the kernels a figure was collected on: sha256 over the sources of libilqg_hip.so (csrc/*.hip, *.hpp in
    name order and the C ABI header), first 16 hex digits.  profiles/traffic.json entries carry the value of the build
    their PMC passes ran on (scripts/summarize_profile.py prints it); a `traffic` figure is only quoted on a match."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ilqgames_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "ilqg.h"), "rb").read())
    return h.hexdigest()[:16]


def sweep_flops_executed_per_step(n, m, N, open_loop):
    """What the device sweep EXECUTES per time step (tile products on 16 x 16 x 4 matrix instructions, 2048 flop each,
    padding included), as opposed to the reference's dense count below.  Open loop (csrc/ilqg_lq_openloop.hpp, n <= 31):
    60 matrix instructions per player wave + the m x m elimination with n + 1 right-hand sides (inversion lemma
    instead of the reference's n x n QR).  Feedback, n <= 16 (csrc/ilqg_lq.hpp): 17 per player wave + the m x m
    elimination."""
    elim = (2.0 / 3.0) * m ** 3 + 2.0 * m * m * (n + 1)
    if open_loop:
        return N * 60 * 2048 + elim
    return N * 17 * 2048 + elim


def sweep_flops_per_step(n, m, N, open_loop):
    """feedback sweep 4Nn^3+4mn^2+2m^2n+2n^2m+(2/3)m^3+2m^2(n+1)+4Nn^2+sum(2m_j^2 n+2m_j n^2);
    open-loop sweep ~ (9N + 4/3) n^3."""
    if open_loop:
        return (9 * N + 4.0 / 3.0) * n ** 3
    mj = m // N
    return (4 * N * n ** 3 + 4 * m * n * n + 2 * m * m * n + 2 * n * n * m + (2.0 / 3.0) * m ** 3 + 2 * m * m * (n + 1) +
            4 * N * n * n + N * (2 * mj * mj * n + 2 * mj * n * n))


class HipBackend:
    """The product path: libilqg_hip.so through ctypes, buffers owned by torch."""

    def __init__(self, spec, dtype, local_rank):
        import torch
        from ilqgames_amd import hip
        self.torch, self.hip = torch, hip
        torch.cuda.set_device(local_rank)
        self.prob = hip.Problem(spec, dtype)
        self.tdtype = hip.torch_dtype(dtype)
        self.group_backend = "nccl"

    def device(self):
        return "cuda"

    def alloc(self, B):
        return self.prob.alloc_solve_buffers(B)

    counted = None  # ilqg_solve_options::counted: None = the library's choice (an asynchronous launch sequence)
    probe_first = 0  # ilqg_solve_options::probe_first: 0 = the library's choice
    single_wave = None  # ilqg_solve_options::single_wave_sweep: None = the library's choice
    split_trial = None  # ilqg_solve_options::split_trial: None = the library's choice
    adjoint = None      # ilqg_solve_options::adjoint_expected_decrease
    static_rows = None  # ilqg_solve_options::static_rows
    padded_sweep = None  # ilqg_solve_options::padded_sweep
    probe_lanes = None  # ilqg_solve_options::probe_lanes

    def solve(self, x0, bufs, iters):
        self.prob.solve(x0, bufs, fixed_iters=iters, counted=self.counted, probe_first=self.probe_first,
                        single_wave_sweep=self.single_wave, split_trial=self.split_trial,
                        adjoint_expected_decrease=self.adjoint, static_rows=self.static_rows,
                        padded_sweep=self.padded_sweep, probe_lanes=self.probe_lanes)

    def sync(self):
        self.torch.cuda.synchronize()

    def events(self):
        return [self.torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def mean_backtracks(self, bufs, iters_total):
        st = self.prob.solve_state(bufs)
        return float(st["backtracks"].sum().item()) / max(1, iters_total)


class StubBackend:
    """CPU stand-in (see the module docstring): the "strategy" of an instance is its global index."""

    def __init__(self, spec, dtype, local_rank):
        import torch
        self.torch = torch
        self.n, self.m, self.N, self.T = spec.n, spec.m, len(spec.subsystems), spec.T
        self.tdtype = torch.float64
        self.group_backend = "gloo"
        self.offset = 0

    def device(self):
        return "cpu"

    def alloc(self, B):
        z = lambda *s: self.torch.zeros(s, dtype=self.tdtype)  # noqa: E731
        return dict(xs=z(B, self.T, self.n), us=z(B, self.T, self.m), P=z(B, self.T, self.m * self.n),
                    alpha=z(B, self.T, self.m), costs=z(B, self.N), iters=self.torch.zeros(B, dtype=self.torch.int32),
                    status=self.torch.zeros(B, dtype=self.torch.int32), converged=self.torch.zeros(B, dtype=self.torch.int32))

    def solve(self, x0, bufs, iters):
        B = x0.shape[0]
        ids = self.torch.arange(self.offset, self.offset + B, dtype=self.tdtype)
        bufs["P"][:] = ids[:, None, None]
        bufs["alpha"][:] = -ids[:, None, None]
        bufs["iters"][:] = iters
        bufs["status"][:] = 1

    def sync(self):
        pass

    def events(self):
        return None

    def mean_backtracks(self, bufs, iters_total):
        return 0.0


def launch_ranks_if_needed(gpus):
    """`--gpus N` is the number of ranks of the job (one process per GPU).  Started by a launcher
    (torch.distributed.run sets WORLD_SIZE) the two must agree — a mismatch is an error, not a silent one-GPU run.
    Started bare with N > 1 (`python bench.py --gpus 8`), this process becomes the launcher: it re-executes itself
    as N ranks under torch.distributed.run on a free local port and exits with the job's status."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != gpus:
            sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks\n" % (gpus, ws))
            sys.exit(2)
        return
    if gpus <= 1:
        return
    import socket
    import subprocess
    rc = 1
    for attempt in range(3):
        # the port is free when probed and re-bound by the launcher a moment later: a job that dies at once (somebody
        # took the port in between) is started again on another one; a job that ran is not
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        t0 = time.time()
        rc = subprocess.call(cmd)
        if rc == 0 or time.time() - t0 > 20.0:
            break
    sys.exit(rc)


HEADLINE_CONFIG = "modified_three_player_intersection"
# shorthands for BASELINE.json's configurations: (config, dtype, instances per GPU, default steps)
BASELINE_CONFIGS = {
    2: (HEADLINE_CONFIG, "f64", 1024, 20),
    3: (HEADLINE_CONFIG, "f32", 8192, 10),
    4: ("roundabout_merging_T150", "f64", 4096, 4),
    5: ("three_player_collision_avoidance_reachability", "f64", 2048, 5),
}


def _bench_spec(examples, config, linesearch="auto"):
    """The workload's problem.  Every configuration runs with ITS OWN solver parameters (the ones its exec/*/main.cpp
    sets; ilqgames_amd/examples.py restates them) except the default one under --linesearch auto: the n = 14 example's
    own line-search fraction (0.9) makes the REFERENCE's line search fail at iteration 2
    (tests/test_gpu_parity.py::test_ilq_solve_free_running...), so its throughput is measured with the line-search
    parameters of exec/three_player_intersection/main.cpp:109-120 (alpha0 = 0.1, fraction 0.001) at a fixed iteration
    count.  --linesearch headline applies those to any configuration, --linesearch own never does.
    Returns (spec, description of the parameters in use)."""
    spec = examples.CONFIGS[config]()
    use_headline = linesearch == "headline" or (linesearch == "auto" and config == HEADLINE_CONFIG)
    if use_headline:
        spec.params.initial_alpha_scaling = 0.1
        spec.params.expected_decrease_fraction = 0.001
        spec.params.max_backtracking_steps = 100
        src = "exec/three_player_intersection/main.cpp:109-120" + (
            "; the n=14 example's own 1.0 / 0.9 fail the reference's line search at iteration 2: own_params"
            if config == HEADLINE_CONFIG else "; --linesearch headline")
    else:
        src = "the example's own parameters (ilqgames_amd/examples.py restates its exec/*/main.cpp)"
    desc = "alpha0=%g frac=%g tol=%g backtracks<=%d (%s)" % (
        spec.params.initial_alpha_scaling, spec.params.expected_decrease_fraction, spec.params.convergence_tolerance,
        spec.params.max_backtracking_steps, src)
    return spec, desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="outer iterations in one timed solve (default 20)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="instances per GPU (default 1024)")
    ap.add_argument("--dtype", choices=["f64", "f32"], default=None)
    ap.add_argument("--config", default=None)
    ap.add_argument("--baseline-config", type=int, choices=sorted(BASELINE_CONFIGS), default=None,
                    help="BASELINE.json configuration number: sets --config / --dtype / --batch (/ --steps) to that "
                         "configuration's per-GPU form, e.g. 3 = fp32, 8192 instances per GPU")
    ap.add_argument("--linesearch", choices=["auto", "own", "headline"], default="auto",
                    help="solver parameters: the configuration's own, or the headline's (see _bench_spec)")
    ap.add_argument("--repeats", type=int, default=5, help="timed repetitions of the K-step solve; the median is reported")
    ap.add_argument("--backend", choices=["hip", "stub"], default="hip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the single-instance ms/solve figure (profiling runs: keeps the kernel statistics to the batch)")
    ap.add_argument("--single-wave", choices=["auto", "on", "off"], default="auto",
                    help="ilqg_solve_options::single_wave_sweep (A/B measurements)")
    ap.add_argument("--adjoint", choices=["auto", "on", "off"], default="auto",
                    help="ilqg_solve_options::adjoint_expected_decrease (A/B measurements)")
    ap.add_argument("--split-trial", choices=["auto", "on", "off"], default="auto",
                    help="ilqg_solve_options::split_trial (A/B measurements)")
    ap.add_argument("--padded-sweep", choices=["auto", "on", "off"], default="auto",
                    help="ilqg_solve_options::padded_sweep (A/B measurements: a run-time-dimensioned solve's sweep on the "
                         "specialised kernel of the shape the game embeds in vs the all-LDS sweep)")
    ap.add_argument("--probe-lanes", choices=["auto", "on", "off"], default="auto",
                    help="ilqg_solve_options::probe_lanes (A/B measurements: probing rollouts with a lane per (candidate, subsystem))")
    ap.add_argument("--static-rows", choices=["auto", "on", "off"], default="auto",
                    help="ilqg_solve_options::static_rows (A/B measurements: straight-line row stage vs the interpreter)")
    ap.add_argument("--probe-first", type=int, default=0,
                    help="ilqg_solve_options::probe_first (A/B measurements of the speculative line search's ramp)")
    ap.add_argument("--no-second-workload", action="store_true",
                    help="skip the back-tracking workload reported beside the headline (three_player_intersection, n = 16)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` block (BASELINE.json configurations 3, 4, 5 in their per-GPU form, ~40 s)")
    ap.add_argument("--no-copy-bandwidth", action="store_true",
                    help="skip the copy-bandwidth measurement behind roofline.peak_measured (profiling runs: keeps the copy "
                         "kernel out of the kernel statistics)")
    ap.add_argument("--cpu-sample", type=int, default=64,
                    help="instances in one run of the CPU baseline sample (64 x 20 iterations ~ 1.3 s on one host thread)")
    args = ap.parse_args()
    base = BASELINE_CONFIGS[args.baseline_config or 2]
    args.config = args.config or base[0]
    args.dtype = args.dtype or base[1]
    args.batch = args.batch or base[2]
    args.steps = args.steps or (base[3] if args.baseline_config else 20)

    launch_ranks_if_needed(args.gpus)

    import torch
    from ilqgames_amd import abi, examples, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    dtype = abi.F64 if args.dtype == "f64" else abi.F32
    elem = 8 if dtype == abi.F64 else 4
    spec, params_desc = _bench_spec(examples, args.config, args.linesearch)
    backend = (HipBackend if args.backend == "hip" else StubBackend)(spec, dtype, local_rank)
    backend.probe_first = args.probe_first
    backend.single_wave = {"auto": None, "on": True, "off": False}[args.single_wave]
    backend.split_trial = {"auto": None, "on": True, "off": False}[args.split_trial]
    backend.adjoint = {"auto": None, "on": True, "off": False}[args.adjoint]
    backend.static_rows = {"auto": None, "on": True, "off": False}[args.static_rows]
    backend.padded_sweep = {"auto": None, "on": True, "off": False}[args.padded_sweep]
    backend.probe_lanes = {"auto": None, "on": True, "off": False}[args.probe_lanes]
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "hip":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    # weak scaling: every rank owns `batch` instances of a global batch of world * batch (contiguous blocks)
    B = args.batch
    total = B * world
    lo, hi = sharding.instance_range(total, rank, world)
    assert hi - lo == B
    x0 = examples.jittered_x0(spec, total, seed=0)[lo:hi] if total <= 65536 else examples.jittered_x0(spec, B, seed=1000003 * rank)
    if args.backend == "stub":
        backend.offset = lo
    x0_d = torch.as_tensor(x0, dtype=backend.tdtype, device=backend.device())
    bufs = backend.alloc(B)

    def reset():
        for k in ("xs", "us", "P", "alpha"):
            bufs[k].zero_()

    def gather():
        # the one exchange step: converged strategies of every instance to rank 0 (sharding.gather_to_root)
        if not distributed:
            return None
        s = torch.cat([bufs["P"].reshape(B, -1), bufs["alpha"].reshape(B, -1)], dim=1)
        return sharding.gather_to_root(s, total, world)

    # warmup (untimed): W iterations
    reset()
    backend.solve(x0_d, bufs, max(1, args.warmup))
    gather()
    backend.sync()
    # A workload whose line searches back-track is run with ilqg_solve_options::counted: the host reads four counters
    # per round and the back-tracking instances finish their line search in the speculative passes
    # instead of serially inside the fused kernel.  Decided from the warm-up's own back-tracking count.
    warm_bt = backend.mean_backtracks(bufs, int(bufs["iters"].sum().item())) if args.backend == "hip" else 0.0
    if args.backend == "hip" and warm_bt > 0.25:
        backend.counted = True

    # timed: exactly K iterations of every instance = K (LQ kernel, trial kernel) rounds after the
    # initial trial pass; all launches are enqueued back to back on the current stream.  The region is repeated
    # `--repeats` times (each bracketed by barrier + synchronize, max over ranks); the MEDIAN repetition is reported.
    reps = []
    for _ in range(max(1, args.repeats)):
        reset()
        if distributed:
            dist.barrier()
        backend.sync()
        ev = backend.events()
        t0 = time.perf_counter()
        if ev:
            ev[0].record()
        backend.solve(x0_d, bufs, args.steps)
        if ev:
            ev[1].record()
        gathered = gather()
        if ev:
            ev[2].record()
        if distributed:
            dist.barrier()
        backend.sync()
        t1 = time.perf_counter()
        el = t1 - t0
        ks = ev[0].elapsed_time(ev[1]) * 1e-3 if ev else el
        gs = ev[1].elapsed_time(ev[2]) * 1e-3 if ev else 0.0
        if distributed:
            tt = torch.tensor([el], dtype=torch.float64, device=backend.device())
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        reps.append((el, ks, gs))
    reps_sorted = sorted(reps)
    elapsed, kernel_s, gather_s = reps_sorted[len(reps_sorted) // 2]

    iters = bufs["iters"].cpu().numpy()
    status = bufs["status"].cpu().numpy()
    # eight instances of the timed batch, kept for the parity figure (BASELINE.json's metric: "P_t rel-err vs CPU")
    sample_ids = sorted(set(int(v) for v in np.linspace(0, B - 1, 8).round()))
    timed_result = {"ids": sample_ids}
    for k in ("P", "alpha", "costs", "xs", "iters"):
        timed_result[k] = bufs[k][sample_ids].cpu().numpy()
    local_iters = int(iters.sum())
    total_iters = local_iters
    if distributed:
        ti = torch.tensor([total_iters], dtype=torch.int64, device=backend.device())
        dist.all_reduce(ti)
        total_iters = int(ti.item())
        assert dist.get_world_size() == world == args.gpus
        if rank == 0:
            # every rank's block arrived: N x batch rows of [P | alpha] on rank 0 (both backends)
            assert gathered is not None and gathered.shape[0] == total == args.gpus * B, (
                "gathered %s rows, expected %d" % (None if gathered is None else gathered.shape[0], total))
            if args.backend == "stub":
                # the stub's strategy of instance b is b: the gather must have put every block where it belongs
                assert torch.equal(gathered[:, 0], torch.arange(total, dtype=gathered.dtype))
            else:
                assert bool(torch.isfinite(gathered).all())

    if rank == 0:
        n, m, N, T = spec.n, spec.m, len(spec.subsystems), spec.T
        pairs = backend.prob.pairs if args.backend == "hip" else [(i, i) for i in range(N)]
        pairs_m = [spec.udims[j] for _, j in pairs]
        mean_bt = backend.mean_backtracks(bufs, local_iters)
        bytes_iter = algorithmic_bytes_per_iteration(n, m, N, T, pairs_m, elem, backtracks=mean_bt)
        bytes_iter_b0 = algorithmic_bytes_per_iteration(n, m, N, T, pairs_m, elem, backtracks=0.0)
        value = total_iters / elapsed
        # Roofline of the hot path on this rank.  One outer iteration of the batch is one round of two
        # kernels (ilq_lq_kernel: Riccati sweep; ilq_trial_kernel: rollout + linearise/quadraticise +
        # line-search decision); the pair is the "launch" the algorithmic bytes are counted for, timed
        # with HIP events over the K rounds (profiles/: the two kernels' rocprofv3 averages add up to it).
        launch_bytes = bytes_iter * local_iters
        achieved = launch_bytes / kernel_s / 1e9
        achieved_b0 = bytes_iter_b0 * local_iters / kernel_s / 1e9
        # HBM traffic per round: bench.py cannot run rocprofv3 on itself, so this figure is READ from the committed
        # medians of the PMC passes scripts/profile.sh collects on this workload (profiles/traffic.json), and labelled so
        # — and only when they were collected on THESE kernels: an entry is stamped with the hash of the kernel sources
        # of the build it was measured on (csrc_sha16), a figure from any other build is not quoted (null)
        traffic, traffic_source = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            ent = tj.get("%s:%s:%d" % (args.config, args.dtype, B))
            sha = csrc_sha16()
            if ent and ent.get("csrc_sha16") == sha:
                traffic = ent["bytes_per_round"]
                traffic_source = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE medians, %s, kernels %s)" % (
                    ent.get("collected", "this round"), sha)
            elif ent:
                traffic_source = "not quoted: profiles/traffic.json holds a figure collected on kernels %s, this build is %s" % (
                    ent.get("csrc_sha16", "(unstamped)"), sha)
        except (OSError, ValueError, KeyError):
            pass
        open_loop = bool(spec.params.open_loop)
        flops_round = sweep_flops_per_step(n, m, N, open_loop) * T * B
        flops_exec_round = sweep_flops_executed_per_step(n, m, N, open_loop) * T * B
        peak_tf = FP64_PEAK_TFLOPS if elem == 8 else FP32_PEAK_TFLOPS
        peak_measured = None
        if args.backend == "hip" and not args.no_copy_bandwidth:
            try:
                peak_measured = backend.hip.copy_bandwidth_gbs()
            except Exception as e:  # the nominal figure stands on its own
                sys.stderr.write("copy-bandwidth measurement skipped: %r\n" % (e,))
        out = {
            "metric": "iLQ iterations/sec (batch)", "value": value, "unit": "instance-iterations/s",
            "n_gpus": dist.get_world_size() if distributed else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s n=%d N=%d T=%d batch=%d/GPU %s, fixed %d outer iterations, %s"
                                   % (args.config, n, N, T, B, args.dtype, args.steps, params_desc),
                       "parallelism": "instances sharded across %d GPU(s); RCCL gather of strategies" % world,
                       "launch_mode": ("host-counted rounds, speculative line search (warm-up back-tracked %.2f times per "
                                       "iteration)" % warm_bt) if getattr(backend, "counted", None) else
                                      "asynchronous launch sequence (no host round trips)"},
            "repeats": {"count": len(reps), "reported": "median", "ms_per_step_all": [r[0] / args.steps * 1e3 for r in reps]},
            "ms_per_solve_batch": kernel_s * 1e3,
            "gather_ms": gather_s * 1e3,
            "success_fraction": float(status.mean()),
            "mean_backtracks": mean_bt,
            # achieved / frac count NO rejected line-search trial (SURVEY.md 8(d)'s b = 0 bytes per accepted iteration):
            # a rejected trial is work the path did, not bytes the workload asked for.  The figure that adds
            # s*T*[W_quad + W_strat + 2 W_op] per rejected trial is kept under `with_rejected_trials`.
            "roofline": {"bound": "hbm", "achieved": achieved_b0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_b0 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         # the nominal 8 TB/s AND the box's measured copy bandwidth as denominators
                         "peak_measured": peak_measured,
                         "frac_of_measured": (achieved_b0 / peak_measured) if peak_measured else None,
                         "peak_measured_source": "ilqg_copy_bandwidth (16 KB contiguous per workgroup, 16 bytes per lane and load, 1 GiB each way, best "
                                                 "of 10 launches, HIP events; read + write bytes counted)",
                         "with_rejected_trials": {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                                                  "mean_backtracks": mean_bt},
                         "kernel": "ilq_lq_kernel + ilq_trial_kernel (one round = one outer iteration of the batch)",
                         "launch_ms": kernel_s * 1e3 / max(1, args.steps),
                         "algorithmic_bytes_per_launch": bytes_iter_b0 * local_iters / max(1, args.steps),
                         "bytes_per_iteration_per_instance": bytes_iter_b0,
                         "frac_b0": achieved_b0 / HBM_PEAK_GBS, "bytes_per_iteration_per_instance_b0": bytes_iter_b0,
                         "flop": {"sweep_tflops": flops_round * args.steps / kernel_s / 1e12,
                                  "sweep_tflops_executed": flops_exec_round * args.steps / kernel_s / 1e12,
                                  "peak_tflops": peak_tf,
                                  "frac_executed": flops_exec_round * args.steps / kernel_s / 1e12 / peak_tf,
                                  "note": "LQ sweep only, over the whole round time: `sweep_tflops` counts the reference's dense "
                                          "flops (SURVEY.md 8d), `sweep_tflops_executed` what the device sweep issues on the matrix "
                                          "cores (16x16x4 tiles, padding included; open loop: inversion lemma instead of the n x n QR); "
                                          "peak = fp32 vector peak of MI355X_MICROARCH.md, halved for fp64 (derived, see bench.py)"}},
        }
        if args.backend == "hip" and world == 1 and not args.no_latency:
            out["latency"] = latency_figures(backend, examples, abi, args, x0_d)
            out["own_params"] = own_params_figure(backend, examples, abi, args, x0_d)
            if args.config == HEADLINE_CONFIG and not args.no_second_workload:
                out["second_workload"] = backtracking_workload(examples, abi, args, local_rank)
            if args.config == HEADLINE_CONFIG and args.baseline_config is None and not args.no_configs:
                out["configs"] = baseline_configs_block(examples, abi, local_rank)
        if args.backend == "hip" and not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, spec, x0, dtype, abi, out.get("latency"))
            out["parity"] = parity_figure(args, spec, x0, dtype, abi, timed_result)
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def latency_figures(backend, examples, abi, args, x0_d):
    """BASELINE.json's second figure, ms per solve: ONE instance run free to its convergence test, zero warm start,
    outside the timed region — on a workload that does converge: the bench's game with alpha0 = 0.5 (fraction 0.001,
    tolerance 1.0): ~400 accepted iterations."""
    import torch
    spec, _ = _bench_spec(examples, args.config, args.linesearch)
    spec.params.initial_alpha_scaling = 0.5
    prob = backend.hip.Problem(spec, abi.F64 if args.dtype == "f64" else abi.F32)
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
    it = int(lb["iters"][0].item())
    out = {"ms_per_solve": sorted(lat)[1] * 1e3, "iterations": it, "ms_per_iteration": sorted(lat)[1] * 1e3 / max(1, it),
           "converged": int(lb["converged"][0].item()), "success": int(lb["status"][0].item()), "instances": 1,
           "mode": "free-running to convergence_tolerance = 1.0, alpha0 = 0.5, fraction 0.001, zero warm start"}
    if args.dtype == "f64":
        # the same solve in the reference's own arithmetic (include/ilqgames/utils/types.h:68-69: float)
        prob32 = backend.hip.Problem(spec, abi.F32)
        x32 = x0_d[:1].float()
        lb32 = prob32.alloc_solve_buffers(1)
        prob32.solve(x32, lb32)
        lat32 = []
        for _ in range(3):
            for k in ("xs", "us", "P", "alpha"):
                lb32[k].zero_()
            torch.cuda.synchronize()
            l0 = time.perf_counter()
            prob32.solve(x32, lb32)
            torch.cuda.synchronize()
            lat32.append(time.perf_counter() - l0)
        it32 = int(lb32["iters"][0].item())
        out["f32"] = {"ms_per_solve": sorted(lat32)[1] * 1e3, "iterations": it32,
                      "ms_per_iteration": sorted(lat32)[1] * 1e3 / max(1, it32),
                      "converged": int(lb32["converged"][0].item()), "success": int(lb32["status"][0].item())}
    return out


def timed_workload(examples, abi, cfg, dtype_name, B, steps, local_rank, warmup=3, reps=5, linesearch="own"):
    """One more fixed-iteration workload timed like the headline (warm-up decides the launch mode, HIP events around
    the solve, inputs resident in HBM, median of `reps`): its own solver parameters unless `linesearch` says otherwise.
    Returns the per-workload block of the bench line."""
    import torch
    spec, params_desc = _bench_spec(examples, cfg, linesearch)
    dtype = abi.F64 if dtype_name == "f64" else abi.F32
    elem = 8 if dtype == abi.F64 else 4
    be = HipBackend(spec, dtype, local_rank)
    x0 = torch.as_tensor(examples.jittered_x0(spec, B, seed=0), dtype=be.tdtype, device="cuda")
    bufs = be.alloc(B)

    def reset():
        for k in ("xs", "us", "P", "alpha"):
            bufs[k].zero_()
    reset()
    be.solve(x0, bufs, warmup)
    be.sync()
    warm_bt = be.mean_backtracks(bufs, int(bufs["iters"].sum().item()))
    if warm_bt > 0.25:
        be.counted = True
    runs = []
    for _ in range(reps):
        reset()
        be.sync()
        ev = be.events()
        ev[0].record()
        be.solve(x0, bufs, steps)
        ev[1].record()
        be.sync()
        runs.append(ev[0].elapsed_time(ev[1]) * 1e-3)
    kernel_s = sorted(runs)[len(runs) // 2]
    iters = int(bufs["iters"].sum().item())
    n, m, N, T = spec.n, spec.m, len(spec.subsystems), spec.T
    pairs_m = [spec.udims[j] for _, j in be.prob.pairs]
    b0 = algorithmic_bytes_per_iteration(n, m, N, T, pairs_m, elem, backtracks=0.0)
    out = {"workload": "%s n=%d N=%d T=%d batch=%d %s%s, fixed %d outer iterations, %s" % (
               cfg, n, N, T, B, dtype_name, " open-loop" if spec.params.open_loop else "", steps, params_desc),
           "value": iters / kernel_s, "unit": "instance-iterations/s", "ms_per_step": kernel_s / steps * 1e3,
           "repeats": len(runs), "ms_per_step_all": [r / steps * 1e3 for r in runs],
           "mean_backtracks": be.mean_backtracks(bufs, iters), "success_fraction": float(bufs["status"].float().mean().item()),
           "launch_mode": "host-counted rounds, speculative line search" if be.counted else "asynchronous launch sequence",
           "roofline": {"bound": "hbm", "achieved": b0 * iters / kernel_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": b0 * iters / kernel_s / 1e9 / HBM_PEAK_GBS, "bytes_per_iteration_per_instance": b0,
                        "note": "b = 0 bytes per accepted iteration (SURVEY.md 8d); rejected trials are not counted as useful bytes"}}
    del bufs, be
    torch.cuda.empty_cache()
    return out


def backtracking_workload(examples, abi, args, local_rank, steps=6, warmup=3):
    """A second reported workload beside the headline, so that the path every line search that back-tracks goes through
    has a driver-visible number: ThreePlayerIntersectionExample (the n = 16 example BASELINE.json's config 2 names, its
    constraints as augmented-Lagrangian terms at their initial multipliers), its own solver parameters
    (exec/three_player_intersection/main.cpp:109-120), the headline's batch and precision, `steps` outer iterations —
    the launch mode chosen from the warm-up exactly as for the headline.  Median of five timed solves."""
    return timed_workload(examples, abi, "three_player_intersection", args.dtype, args.batch, steps, local_rank, warmup=warmup)


def receding_horizon_workload(examples, abi, local_rank, batch=2048, replans=12, horizon_calls=200, max_solver_iters=50,
                              dtype_name="f64", cpu=True):
    """BASELINE.json config 5 AS WRITTEN: the receding-horizon loop of src/receding_horizon_simulator.cpp:65-137 over
    three_player_collision_avoidance_reachability with the augmented-Lagrangian solver, `batch` plans resident on the
    device, every call after the first warm-started from the spliced plan.  The wall clock of the reference's loop is
    replaced by a fixed simulated solve time sized so that `horizon_calls` (200) calls fit the horizon; the first
    1 + `replans` of them are run and timed (a 200-call run is minutes; what a call costs does not change along it).
    An instance whose first solve fails leaves the loop there, as the reference's CHECK(success) (:77) ends its run —
    `active_after_first_call` says how many plans the replanning calls are really about.
    `cpu`: the CPU port (the oracle's restatement of the same loop) on the plans that survived the device's first call
    — at most 64 of them — with one thread and with every usable core, first call and replans timed apart (two runs:
    one record, then 1 + replans records), so that the device's replanning rate has its CPU figure beside it."""
    import torch
    cfg = "three_player_collision_avoidance_reachability"
    spec = examples.CONFIGS[cfg]()
    spec.params.max_solver_iters = max_solver_iters
    from ilqgames_amd import hip
    torch.cuda.set_device(local_rank)
    dtype = abi.F64 if dtype_name == "f64" else abi.F32
    prob = hip.Problem(spec, dtype)
    x0 = examples.jittered_x0(spec, batch, seed=1)
    tick = 0.5 * spec.T * spec.dt / (horizon_calls + 8)
    stamps, active, iterates, active_masks = [], [], [], []

    def on_record(r, info):
        torch.cuda.synchronize()
        stamps.append(time.perf_counter())
        active.append(int(info["active"].sum().item()))
        iterates.append(int((info["bufs"]["iters"] * info["active"]).sum().item()))
        if r <= 1:
            active_masks.append(info["active"].cpu().numpy().astype(bool))

    # untimed warm-up, as for every other workload: the first call and one replan.  The loop around the solver calls is
    # this harness's (torch.where / sum / mask updates on device tensors), and the first use of each of those torch
    # kernels loads its code object: 55 + 100 + 68 ms of idle device in the first replan of a traced process
    # (scripts/trace_cmd.sh), 75-270 ms un-traced — not something a 12-replan average of the library should carry.
    prob.receding_horizon_simulate(x0, final_time=1e9, planner_runtime=tick, extra_time=tick, solve_time=tick,
                                   augmented_lagrangian=True, max_records=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = prob.receding_horizon_simulate(x0, final_time=1e9, planner_runtime=tick, extra_time=tick, solve_time=tick,
                                         augmented_lagrangian=True, max_records=replans + 1, on_record=on_record)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    solves = int(out["num_records"].sum().item())
    first_s = stamps[0] - t0 if stamps else wall
    replan_s = (stamps[-1] - stamps[0]) if len(stamps) > 1 else 0.0
    replan_solves = sum(active[1:])
    res = {"workload": "%s n=%d N=%d T=%d batch=%d %s: RecedingHorizonSimulator loop, AugmentedLagrangianSolver, first call + %d "
                       "warm-started replans of a %d-call horizon (simulated solve time %.4f s), max_solver_iters=%d"
                       % (cfg, spec.n, len(spec.subsystems), spec.T, batch, dtype_name, out["calls"] - 1, horizon_calls, tick,
                          max_solver_iters),
           "dtype": dtype_name,
           "calls": out["calls"], "instance_solves": solves, "seconds": wall,
           "instance_solves_per_s": solves / wall, "unit": "warm-started instance-solves/s (first call included)",
           "first_call_ms": first_s * 1e3, "ms_per_replan": (replan_s / max(1, out["calls"] - 1)) * 1e3,
           "replan_instance_solves_per_s": (replan_solves / replan_s) if replan_s > 0 else None,
           "replan_logged_iterates_per_s": (sum(iterates[1:]) / replan_s) if replan_s > 0 else None,
           "active_after_first_call": active[1] if len(active) > 1 else (active[0] if active else 0),
           "active_at_end": int(out["active"].sum().item()),
           "logged_iterates_per_call": iterates,
           "ms_per_call": [round((b - a) * 1e3, 2) for a, b in zip([t0] + stamps[:-1], stamps)]}
    del prob
    torch.cuda.empty_cache()
    if cpu and len(active_masks) > 1 and active_masks[1].any():
        res["cpu_port"] = receding_horizon_cpu(spec, abi, dtype, x0[active_masks[1]][:64], tick, replans)
    return res


def receding_horizon_cpu(spec, abi, dtype, x0_all, tick, replans):
    """The CPU port of the same loop (oracle SimulateBatch: RecedingHorizonSimulator + AugmentedLagrangianSolver +
    SolutionSplicer, same simulated solve time) on the given plans: one thread, then every usable core (OpenMP over the
    plans).  Replanning time = (run with 1 + replans records) - (run with the first call only)."""
    from oracle import pyoracle
    op = pyoracle.OracleProblem(spec)
    ncpu = usable_cpus()
    out = {}
    # bounded: a replanning call of ONE plan is ~0.3 s of CPU here (its failing line searches walk through all their
    # step sizes, each a rollout and a quadraticisation), so one thread gets three plans and every core one plan each
    for label, threads, plans in (("1_thread", 1, 3), ("all_cores", ncpu, ncpu)):
        x0 = x0_all[:plans]
        c0 = time.perf_counter()
        op.receding_horizon_simulate(dtype, x0, 1e9, tick, extra_time=tick, solve_time=tick, augmented_lagrangian=True,
                                     max_records=1, threads=threads)
        c1 = time.perf_counter()
        full = op.receding_horizon_simulate(dtype, x0, 1e9, tick, extra_time=tick, solve_time=tick, augmented_lagrangian=True,
                                            max_records=replans + 1, threads=threads)
        c2 = time.perf_counter()
        rs = max(1e-9, (c2 - c1) - (c1 - c0))
        nrec = full["num_records"]
        out[label] = {"threads": threads, "plans": int(x0.shape[0]), "first_call_ms": (c1 - c0) * 1e3, "replans_s": rs,
                      "ms_per_replan": rs / max(1, replans) * 1e3,
                      "replan_instance_solves": int((nrec - 1).clip(min=0).sum()),
                      "replan_instance_solves_per_s": float((nrec - 1).clip(min=0).sum()) / rs,
                      "replan_logged_iterates_per_s": float(sum(int(full["iters"][b, 1:int(nrec[b])].sum()) for b in range(len(nrec)))) / rs}
    return out


def baseline_configs_block(examples, abi, local_rank):
    """Every other BASELINE.json configuration in its per-GPU form, in the default bench line (`configs`)."""
    out = {}
    # config 3: fp32, 8192 instances per GPU (65536 over 8)
    out["config3_f32_b8192"] = timed_workload(examples, abi, HEADLINE_CONFIG, "f32", 8192, 10, local_rank, linesearch="auto")
    # the headline batch in fp32 and the headline shape at the large batch (the single-wave throughput schedule)
    out["headline_f32_b1024"] = timed_workload(examples, abi, HEADLINE_CONFIG, "f32", 1024, 20, local_rank, linesearch="auto")
    out["headline_f64_b8192"] = timed_workload(examples, abi, HEADLINE_CONFIG, "f64", 8192, 10, local_rank, linesearch="auto")
    # config 4: roundabout merging, n = 24, N = 4, T = 150, open-loop sweep, 4096 instances, its own parameters
    out["config4_roundabout_T150_openloop_b4096"] = timed_workload(examples, abi, "roundabout_merging_T150", "f64", 4096, 4, local_rank)
    # config 5's scene as a fixed-iteration solve (what --baseline-config 5 times), then config 5 as written
    out["config5_scene_fixed_iterations_b2048"] = timed_workload(examples, abi, "three_player_collision_avoidance_reachability",
                                                                 "f64", 2048, 5, local_rank)
    # a game whose dimensions have no instantiation (a test scene, not a reference example: Dubins car m = 1 beside
    # Car5D m = 2, n = 8): the run-time-dimensioned kernels, its sweep on the padded (8, 2, 2) kernel
    out["mixed_dubins_car_scene_b1024"] = timed_workload(examples, abi, "mixed_dubins_car_scene", "f64", 1024, 10, local_rank)
    out["config5_receding_horizon_al_b2048"] = receding_horizon_workload(examples, abi, local_rank)
    # ... and in fp32, the reference's own arithmetic (include/ilqgames/utils/types.h:68-69), in which its x0 passes
    out["config5_receding_horizon_al_b2048_f32"] = receding_horizon_workload(examples, abi, local_rank, dtype_name="f32")
    return out


def own_params_figure(backend, examples, abi, args, x0_d):
    """The example's OWN solver parameters (exec/modified_three_player_intersection_example/main.cpp:74-76,110-116 for
    the default config), free-running on the whole batch: what a user of the unmodified example gets."""
    import torch
    spec = examples.CONFIGS[args.config]()
    prob = backend.hip.Problem(spec, abi.F64 if args.dtype == "f64" else abi.F32)
    B = x0_d.shape[0]
    bufs = prob.alloc_solve_buffers(B)
    prob.solve(x0_d, bufs)
    for k in ("xs", "us", "P", "alpha"):
        bufs[k].zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prob.solve(x0_d, bufs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    it = bufs["iters"].cpu().numpy()
    return {"ms_per_solve_batch": dt * 1e3, "mean_iterations": float(it.mean()), "success_fraction": float(bufs["status"].float().mean().item()),
            "instance_iterations_per_s": float(it.sum()) / dt,
            "params": "alpha0=%g frac=%g tol=%g" % (spec.params.initial_alpha_scaling, spec.params.expected_decrease_fraction,
                                                   spec.params.convergence_tolerance)}


def parity_figure(args, spec, x0, dtype, abi, timed):
    """BASELINE.json's metric names "P_t rel-err vs CPU": the strategies, trajectory and total costs the TIMED batch ended
    with (eight instances spread over it, after the `steps` fixed iterations just timed, line searches included) against
    the CPU oracle run on the same initial states in the same precision.  The oracle is the checker here, nothing else."""
    from oracle import pyoracle
    ids = timed["ids"]
    ref = pyoracle.OracleProblem(spec).solve(dtype, x0[ids], fixed_iters=args.steps, threads=1)

    def rel(a, b):
        return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / max(1e-300, np.max(np.abs(b))))
    same = [int(i) for i in range(len(ids)) if int(timed["iters"][i]) == int(ref["iters"][i])]
    fig = {"instances": [int(i) for i in ids], "iterations": args.steps, "dtype": args.dtype,
           "instances_with_equal_iteration_counts": len(same),
           "against": "oracle/ilqg_oracle.hpp (CPU restatement of the reference, same precision, same x0, same fixed iteration count)"}
    if same:
        fig["rel_err_P"] = max(rel(timed["P"][i], ref["P"][i]) for i in same)
        fig["rel_err_alpha"] = max(rel(timed["alpha"][i], ref["alpha"][i]) for i in same)
        fig["rel_err_total_cost"] = max(rel(timed["costs"][i], ref["costs"][i]) for i in same)
        fig["rel_err_xs"] = max(rel(timed["xs"][i], ref["xs"][i]) for i in same)
        fig["max"] = max(fig["rel_err_P"], fig["rel_err_alpha"], fig["rel_err_total_cost"])
    return fig


def usable_cpus():
    """CPUs this process can really use: the affinity mask, cut by the cgroup CPU quota if there is one (a container
    that shows 256 CPUs but is throttled to 16 would otherwise run 256 threads on 16 cores' worth of time)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(args, spec, x0, dtype, abi, latency):
    """The oracle (a port: the reference binary cannot be built here, Eigen3 / glog / gflags absent) timed on this
    box's host cores on a bounded sample of the same workload: one thread — the reference's execution model — as the
    median of five runs in fp32 (the reference's arithmetic) and in fp64, and every core with OpenMP over the
    instances (oracle/oracle_capi.cpp, dynamic schedule), sized for >= 1 s of work per thread."""
    from ilqgames_amd import examples
    from oracle import pyoracle
    S = min(args.cpu_sample, x0.shape[0])
    op = pyoracle.OracleProblem(spec)
    res = {}
    for name, dt_ in (("f32", abi.F32), ("f64", abi.F64)):
        op.solve(dt_, x0[:2], fixed_iters=2)  # warm the code path
        runs = []
        for _ in range(5):
            c0 = time.perf_counter()
            ref = op.solve(dt_, x0[:S], fixed_iters=args.steps, threads=1)
            runs.append(int(ref["iters"].sum()) / (time.perf_counter() - c0))
        res[name] = sorted(runs)[2]
    ncpu = usable_cpus()
    # every usable core: instances per thread = what one thread does in ~1.2 s at the rate just measured
    per = max(4, int(1.2 * res[args.dtype] / max(1, args.steps)) + 1)
    n_all = min(ncpu * per, 65536)
    all_cores, all_wall = None, None
    try:
        xa = examples.jittered_x0(spec, n_all, seed=17)
        op.solve(dtype, xa[:ncpu], fixed_iters=1, threads=ncpu)  # spins the thread team up
        runs = []
        for _ in range(3):
            c0 = time.perf_counter()
            ref = op.solve(dtype, xa, fixed_iters=args.steps, threads=ncpu)
            w_ = time.perf_counter() - c0
            runs.append((int(ref["iters"].sum()) / w_, w_))
            if sum(r[1] for r in runs) > 8.0:  # bounded: a box that scales badly does not get three slow runs
                break
        all_cores, all_wall = sorted(runs)[len(runs) // 2]
    except Exception as e:  # the single-thread figure stands on its own
        sys.stderr.write("all-core CPU baseline skipped: %r\n" % (e,))
    out = {"value": res[args.dtype], "unit": "instance-iterations/s", "cores": 1, "kind": "port",
           "value_f32": res["f32"], "value_f64": res["f64"],
           "sample": "median of 5 runs of %d of the same instances x %d iterations, oracle/ilqg_oracle.hpp, 1 thread — the "
                     "reference's execution model; the reference binary itself cannot be built here (Eigen3/glog/gflags absent)"
                     % (S, args.steps),
           "value_all_cores": all_cores, "cores_all": ncpu,
           "cores_visible": os.cpu_count(),
           "sample_all_cores": "median run: OpenMP over %d instances (%d per thread) x %d iterations on %d threads (the CPUs this "
                               "process may use: affinity mask and cgroup quota; os.cpu_count() = %s), %s, %.2f s per run"
                               % (n_all, per, args.steps, ncpu, os.cpu_count(), args.dtype, all_wall or 0.0)}
    if latency is not None:
        s2, _ = _bench_spec(examples, args.config, args.linesearch)
        s2.params.initial_alpha_scaling = 0.5
        c0 = time.perf_counter()
        one = pyoracle.OracleProblem(s2).solve(dtype, x0[:1])
        latency["cpu_ms_per_solve"] = (time.perf_counter() - c0) * 1e3
        latency["cpu_iterations"] = int(one["iters"][0])
    return out


if __name__ == "__main__":
    main()
