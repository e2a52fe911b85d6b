"""The statically specialised row stage (csrc/ilqg_rows.hpp: ProgStatic<ID>, csrc/ilqg_rowprog_static.hpp): a problem whose
row program matches a registered structure runs ComputeLinearization / ComputeCostQuadraticization / the merit and cost
pieces (src/ilq_solver.cpp:400-490, src/player_cost.cpp:194-225) as straight-line code instead of interpreting the
program — the same expressions in the same order, so the same bits."""
import numpy as np
import pytest

from ilqgames_amd import abi, examples


def _np(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from ilqgames_amd import hip as h
    return h


REGISTERED = ["modified_three_player_intersection", "three_player_intersection",
              "three_player_collision_avoidance_reachability", "roundabout_merging"]


@pytest.mark.parametrize("scene", REGISTERED)
def test_registered_structures_match_the_committed_table(scene):
    """CPU: the program the library builds for each registered scene (ilqg_row_program_build, host only) matches its
    entry of the committed csrc/ilqg_rowprog_static.hpp — a change to build_row_program or to a scene without
    `python scripts/gen_static_rowprogs.py` would silently put these workloads back on the interpreter."""
    from ilqgames_amd import hip
    spec = examples.CONFIGS[scene]()
    words, ident = hip.row_program_build(spec, abi.F64)
    assert ident == REGISTERED.index(scene) + 1
    assert words[12] == len(words)  # RP_WORDS
    # other weights, nominal values, regularisation and lane geometry: the same structure, the same code
    for t in spec.terms:
        t["weight"] = float(t["weight"]) * 1.5 + 0.25 if "weight" in t else 0.0
    _, again = hip.row_program_build(spec, abi.F32)
    assert again == ident
    # another structure (the first term gated like a FinalTimeCost) is not this one
    spec2 = examples.CONFIGS[scene]()
    spec2.terms[0]["first_step"] = 3
    _, other = hip.row_program_build(spec2, abi.F64)
    assert other != ident


def test_unregistered_structure_runs_the_interpreter():
    from ilqgames_amd import hip
    _, ident = hip.row_program_build(examples.CONFIGS["skeleton"](), abi.F64)
    assert ident == 0


@pytest.mark.gpu
@pytest.mark.parametrize("scene", REGISTERED)
@pytest.mark.parametrize("dtype", [abi.F64, abi.F32])
def test_static_rows_are_the_interpreter_bit_for_bit(hip, scene, dtype):
    """Free-running solves (line searches, hand-off to the split passes included) with the row stage as straight-line
    code and as the interpreter: every output identical, and the schedule report says which ran."""
    import torch
    spec = examples.CONFIGS[scene]()
    spec.params.max_solver_iters = 8
    B = 24
    x0 = examples.jittered_x0(spec, B, seed=11)
    prob = hip.Problem(spec, dtype)
    assert prob.row_program()[1] == REGISTERED.index(scene) + 1
    al = spec.num_constraints > 0
    # n = 24 keeps its state rows in an LDS image (more than 16 states): its fused kernel interprets, the static code is
    # in the split row kernels — which is what its full-size batches run (BASELINE config 4)
    kw = dict(split_trial=True) if spec.n > 16 else {}
    a = prob.solve(x0, augmented_lagrangian=al, static_rows=True, **kw)
    torch.cuda.synchronize()
    assert prob.last_schedule() & abi.SCHEDULE_STATIC_ROWS
    a = {q: _np(a[q]).copy() for q in ("xs", "us", "P", "alpha", "costs", "iters", "status", "converged")}
    b = prob.solve(x0, augmented_lagrangian=al, static_rows=False, **kw)
    torch.cuda.synchronize()
    assert not prob.last_schedule() & abi.SCHEDULE_STATIC_ROWS
    for q, v in a.items():
        assert np.array_equal(v, _np(b[q])), q
    assert a["iters"].max() >= 1


@pytest.mark.gpu
def test_static_rows_fixed_iterations_headline_batch(hip):
    """The headline's own launch sequence (fixed iterations, asynchronous) at a batch of a few hundred."""
    import torch
    spec = examples.modified_three_player_intersection()
    spec.params.initial_alpha_scaling = 0.1
    spec.params.expected_decrease_fraction = 0.001
    x0 = examples.jittered_x0(spec, 300, seed=2)
    prob = hip.Problem(spec, abi.F64)
    a = prob.solve(x0, fixed_iters=4)
    torch.cuda.synchronize()
    assert prob.last_schedule() & abi.SCHEDULE_STATIC_ROWS  # AUTO: on
    a = {q: _np(a[q]).copy() for q in ("xs", "us", "P", "alpha", "costs")}
    b = prob.solve(x0, fixed_iters=4, static_rows=False)
    for q, v in a.items():
        assert np.array_equal(v, _np(b[q])), q
