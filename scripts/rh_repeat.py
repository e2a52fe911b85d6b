"""bench.py's config-5-as-written leg alone, several times in one process (diagnostic): per-call times of every run.
   python scripts/rh_repeat.py [runs] [f64|f32]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ilqgames_amd import abi, examples  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dt = sys.argv[2] if len(sys.argv) > 2 else "f64"
for r in range(runs):
    out = bench.receding_horizon_workload(examples, abi, 0, dtype_name=dt, cpu=False)
    print(r, "first %.0f ms, per replan %.1f ms:" % (out["first_call_ms"], out["ms_per_replan"]), out["ms_per_call"][1:], flush=True)
