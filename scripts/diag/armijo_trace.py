"""Which Armijo test fails, and by how much, on the reference examples' OWN initial states and parameters (no jitter).

The oracle (oracle/ilqg_oracle.hpp: SolveILQ, a restatement of src/ilq_solver.cpp:76-172 with ModifyLQStrategies
:289-348 inlined) records every CheckArmijoCondition call (:350-362):
    accept  <=>  last_merit - merit >= expected_decrease_fraction * step * expected_decrease
Run for the two scenes whose own parameters make most line searches fail:
  * ModifiedThreePlayerIntersectionExample, alpha0 = 1.0, fraction 0.9  (exec/modified_three_player_intersection_example/main.cpp:74-76)
  * ThreePlayerCollisionAvoidanceReachabilityExample, alpha0 = 0.1, fraction 0.1
    (exec/receding_horizon_three_player_collision_avoidance_reachability_example/main.cpp:74-81)
in fp32 (the reference's arithmetic, types.h:68-69) and fp64.  CPU only:  python scripts/diag/armijo_trace.py [--markdown]"""
import argparse
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np  # noqa: E402
from ilqgames_amd import abi, examples  # noqa: E402
from oracle import pyoracle  # noqa: E402


def trace(spec, dtype, x0, max_entries=4096):
    op = pyoracle.OracleProblem(spec)
    out = np.zeros((max_entries, 8))
    ok, it = C.c_int(0), C.c_int(0)
    x = np.ascontiguousarray(x0, dtype=np.float32 if dtype == abi.F32 else np.float64)
    lib = pyoracle.lib()
    lib.oracle_armijo_trace.restype = C.c_int
    n = lib.oracle_armijo_trace(op.h, dtype, x.ctypes.data_as(C.c_void_p), max_entries, out.ctypes.data_as(C.POINTER(C.c_double)),
                                C.byref(ok), C.byref(it))
    return out[:min(n, max_entries)], ok.value, it.value, n


def report(name, spec, md):
    x0 = np.asarray(spec.x0, dtype=np.float64)  # the example's own initial state, no jitter
    for dname, dtype in (("fp32", abi.F32), ("fp64", abi.F64)):
        tr, ok, it, n = trace(spec, dtype, x0)
        print("\n%s%s, %s: success = %d after %d iterations, %d Armijo tests (alpha0 = %g, fraction = %g, max back-tracks = %d)" % (
            "### " if md else "", name, dname, ok, it, n, spec.params.initial_alpha_scaling, spec.params.expected_decrease_fraction,
            spec.params.max_backtracking_steps))
        hdr = ("iteration", "back-track", "step", "last merit", "trial merit", "decrease", "fraction x step x ED", "ED", "accepted")
        rows = []
        last_it = None
        shown = 0
        for e in tr:
            i, bt, acc = int(e[0]), int(e[1]), int(e[2])
            # every accepted test, and of a failing search the first three, powers of two after, and the last
            show = acc or bt < 3 or (bt & (bt - 1)) == 0 or bt == spec.params.max_backtracking_steps - 1
            if not show:
                continue
            rows.append((i, bt, "%.3g" % e[3], "%.9g" % e[4], "%.9g" % e[5], "%.3g" % (e[4] - e[5]), "%.3g" % e[7], "%.6g" % e[6],
                         "yes" if acc else "no"))
        if md:
            print("| " + " | ".join(hdr) + " |")
            print("|" + "---|" * len(hdr))
            for r in rows:
                print("| " + " | ".join(str(v) for v in r) + " |")
        else:
            print("  ".join("%-12s" % h for h in hdr))
            for r in rows:
                print("  ".join("%-12s" % str(v) for v in r))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--markdown", action="store_true")
    a = ap.parse_args()
    report("ModifiedThreePlayerIntersectionExample (n = 14)", examples.modified_three_player_intersection(), a.markdown)
    report("ThreePlayerCollisionAvoidanceReachabilityExample (n = 15, config 5's scene)",
           examples.three_player_collision_avoidance_reachability(), a.markdown)


if __name__ == "__main__":
    main()
