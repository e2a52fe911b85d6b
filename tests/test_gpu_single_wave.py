"""GPU parity of the single-wave feedback sweep (ilqgames_amd/csrc/ilqg_lq_feedback1w.hpp): the throughput form the
library picks for batches of five or more instances per CU, forced here onto small batches
(ilqg_solve_options::single_wave_sweep = ON).  Same recursion and the same order of operations per player as the
player-parallel sweep, so: against the oracle after every forced-step iteration (1e-9 fp64, fp32 tolerances of
test_gpu_forced.py), against the player-parallel sweep itself on a free-running solve (same line-search decisions), and
at BASELINE config 3's per-GPU batch through batch independence (a slice solved alone reproduces its rows of the batch
bit for bit)."""
import numpy as np
import pytest

from ilqgames_amd import abi, examples
from helpers import rel_err
from test_gpu_generic import _compare_forced, _forced, _np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from ilqgames_amd import hip as h
    return h


SCENES = ["modified_three_player_intersection",             # (14, 3, 2): the headline
          "three_player_collision_avoidance_reachability",  # (15, 3, 2): max-over-time player
          "two_player_unicycle_4d_scene",                   # (4, 2, 2): shared state
          "dubins_origin"]                                  # (6, 2, 1): one control per player


@pytest.mark.parametrize("scene", SCENES)
@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
def test_single_wave_sweep_matches_oracle_after_every_forced_iteration(hip, oracle, scene, dtype):
    spec = examples.CONFIGS[scene]()
    K, B = 5, 10
    x0, op, steps, x0n = _forced(oracle, spec, B, K, seed=41)
    _compare_forced(hip, op, spec, dtype, x0, steps, x0n, K, dict(single_wave_sweep=True), min_cover=0.6)


@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
def test_single_wave_and_player_parallel_sweeps_agree_on_a_free_running_solve(hip, dtype):
    """Two device schedules of one sweep: the strategies of every iteration agree to rounding, so a free-running solve
    takes the same line-search decisions (iteration counts, success, convergence) wherever those are not within rounding
    of their thresholds — asserted for at least three quarters of the instances — and ends at the same iterate."""
    spec = examples.modified_three_player_intersection()
    spec.params.initial_alpha_scaling = 0.1
    spec.params.expected_decrease_fraction = 0.001
    spec.params.max_backtracking_steps = 100
    spec.params.max_solver_iters = 25
    B = 48
    x0 = examples.jittered_x0(spec, B, seed=3)
    prob = hip.Problem(spec, dtype)
    a = prob.solve(x0, single_wave_sweep=False)
    b = prob.solve(x0, single_wave_sweep=True)
    same = (_np(a["iters"]) == _np(b["iters"])) & (_np(a["status"]) == _np(b["status"])) & \
        (_np(a["converged"]) == _np(b["converged"]))
    assert same.mean() >= 0.75, same
    tol = 1e-7 if dtype == abi.F64 else 5e-2
    for i in np.nonzero(same)[0]:
        assert rel_err(_np(a["xs"])[i], _np(b["xs"])[i]) < tol, i


def test_single_wave_sweep_at_config3_batch_is_batch_independent(hip):
    """BASELINE config 3's per-GPU share (fp32, 8192 instances: the batch size the library picks this sweep at by
    itself): a 37-instance slice solved alone (still on the single-wave sweep) reproduces its rows of the full batch
    bit for bit — instances share nothing, whatever the residency."""
    spec = examples.modified_three_player_intersection()
    spec.params.initial_alpha_scaling = 0.1
    spec.params.expected_decrease_fraction = 0.001
    spec.params.max_backtracking_steps = 100
    B = 8192
    x0 = examples.jittered_x0(spec, B, seed=0)
    prob = hip.Problem(spec, abi.F32)
    full = prob.solve(x0, fixed_iters=3)
    lo = 4111
    part = prob.solve(x0[lo:lo + 37], fixed_iters=3, single_wave_sweep=True)
    for q in ("xs", "us", "P", "alpha", "costs"):
        assert np.array_equal(_np(full[q])[lo:lo + 37], _np(part[q])), q
    # and the two sweeps agree at that size to fp32 rounding through three iterations
    pw = prob.solve(x0[lo:lo + 37], fixed_iters=3, single_wave_sweep=False)
    assert rel_err(_np(pw["xs"]), _np(part["xs"])) < 2e-3


def test_last_schedule_reports_what_auto_chose(hip):
    """ilqg_problem_last_schedule: AUTO picks the single-wave sweep (with the adjoint expected decrease and the split
    trial pass) from five instances per CU on, the player-parallel sweep below; pinned choices are reported as pinned."""
    import torch
    spec = examples.modified_three_player_intersection()
    spec.params.expected_decrease_fraction = 0.001
    spec.params.initial_alpha_scaling = 0.1
    prob = hip.Problem(spec, abi.F64)
    prob.solve(examples.jittered_x0(spec, 8, seed=1), fixed_iters=1)
    torch.cuda.synchronize()
    few = prob.last_schedule()
    assert few & abi.SCHEDULE_COMPACT_ROWS and not few & abi.SCHEDULE_SINGLE_WAVE_SWEEP and not few & abi.SCHEDULE_OPEN_LOOP
    cus = hip.device_info()[1]
    prob.solve(examples.jittered_x0(spec, 5 * cus, seed=1), fixed_iters=1)
    torch.cuda.synchronize()
    many = prob.last_schedule()
    assert many & abi.SCHEDULE_SINGLE_WAVE_SWEEP and many & abi.SCHEDULE_ADJOINT_DECREASE and many & abi.SCHEDULE_SPLIT_TRIAL
    prob.solve(examples.jittered_x0(spec, 5 * cus, seed=1), fixed_iters=1, single_wave_sweep=False)
    torch.cuda.synchronize()
    assert not prob.last_schedule() & abi.SCHEDULE_SINGLE_WAVE_SWEEP


@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
def test_a_masked_batch_is_scheduled_by_the_instances_that_take_part(hip, dtype):
    """A free-running solve under ilqg_solve_options::active counts its mask before it chooses a schedule (the receding-
    horizon loop replans the few plans still running of a large batch, src/receding_horizon_simulator.cpp:77): 40
    instances of a buffer of five per CU run the schedule of a batch of 40 — no single-wave sweep, no split trial pass —
    and return the bits of those 40 solved alone; the other rows are left as they were."""
    import torch
    spec = examples.modified_three_player_intersection()
    spec.params.initial_alpha_scaling = 0.1
    spec.params.expected_decrease_fraction = 0.001
    spec.params.max_backtracking_steps = 100
    spec.params.max_solver_iters = 6
    cus = hip.device_info()[1]
    B = 5 * cus
    x0 = examples.jittered_x0(spec, B, seed=9)
    prob = hip.Problem(spec, dtype)
    rows = np.arange(17, B, B // 40)[:40]
    mask = np.zeros(B, np.int32)
    mask[rows] = 1
    bufs = prob.alloc_solve_buffers(B)
    skipped = np.nonzero(mask == 0)[0]
    for q in ("xs", "us", "P", "alpha", "costs"):
        bufs[q][torch.from_numpy(skipped).cuda()] = 7.0  # (the rows taking part keep the zero warm start)
    out = prob.solve(x0, bufs=bufs, active=torch.from_numpy(mask).cuda())
    torch.cuda.synchronize()
    sched = prob.last_schedule()
    assert not sched & abi.SCHEDULE_SINGLE_WAVE_SWEEP and not sched & abi.SCHEDULE_SPLIT_TRIAL
    alone = prob.solve(x0[rows])
    torch.cuda.synchronize()
    assert prob.last_schedule() == sched
    for q in ("xs", "us", "P", "alpha", "costs", "iters", "status", "converged"):
        assert np.array_equal(_np(out[q])[rows], _np(alone[q])), q
    for q in ("xs", "us", "P", "alpha", "costs"):
        assert np.all(_np(out[q])[skipped] == 7.0), q
    # a fixed-iteration solve stays asynchronous: it keeps the schedule of the buffer's length
    prob.solve(x0, bufs=bufs, active=torch.from_numpy(mask).cuda(), fixed_iters=1)
    torch.cuda.synchronize()
    assert prob.last_schedule() & abi.SCHEDULE_SINGLE_WAVE_SWEEP


@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
def test_deterministic_option_gives_the_same_bits_at_every_batch_size(hip, dtype):
    """ilqg_solve_options::deterministic: no scheduling choice depends on the batch size, so the SAME 64 initial states
    return the same bits solved alone, as rows of a 1024-instance batch and as rows of an 8192-instance batch (where
    AUTO alone would switch to the single-wave sweep and its adjoint expected decrease) — free-running solves, line
    searches included."""
    import torch
    spec = examples.modified_three_player_intersection()
    spec.params.initial_alpha_scaling = 0.1
    spec.params.expected_decrease_fraction = 0.001
    spec.params.max_backtracking_steps = 100
    spec.params.max_solver_iters = 6
    x0 = examples.jittered_x0(spec, 8192, seed=5)
    prob = hip.Problem(spec, dtype)
    results = {}
    for B in (64, 1024, 8192):
        lo = 0 if B == 64 else 517
        xb = x0[:B].copy()
        xb[lo:lo + 64] = x0[:64]
        out = prob.solve(xb, deterministic=True)
        torch.cuda.synchronize()
        assert not prob.last_schedule() & abi.SCHEDULE_SINGLE_WAVE_SWEEP
        results[B] = {q: _np(out[q])[lo:lo + 64].copy() for q in ("xs", "us", "P", "alpha", "costs", "iters", "status", "converged")}
    for B in (1024, 8192):
        for q, v in results[64].items():
            assert np.array_equal(v, results[B][q]), (B, q)
    # without the option the large batch runs the other sweep (the reason the option exists)
    prob.solve(x0, fixed_iters=1)
    torch.cuda.synchronize()
    assert prob.last_schedule() & abi.SCHEDULE_SINGLE_WAVE_SWEEP
