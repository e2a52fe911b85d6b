"""Writes tests/golden/forced_coverage.json: how many (instance, iteration) pairs tests/test_gpu_forced.py compares per
scene and dtype.  The mask is a function of ORACLE runs alone (amplification of a 1e-12 nudge in the fp64 oracle; for
fp32, how far the fp32 oracle is from the fp64 one), so it is computed here on the CPU and committed; the GPU test fails
if its own count falls below it.  Run from the repo root:  python tests/golden/make_forced_coverage.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ilqgames_amd import abi  # noqa: E402
from oracle import pyoracle  # noqa: E402
import test_gpu_forced as t  # noqa: E402

pyoracle.build(force=False)
out = {}
for scene in t.SCENES:
    spec, B, x0, op, steps, x0_nudged = t.forced_case(pyoracle, scene)
    for dtype, name in ((abi.F64, "f64"), (abi.F32, "f32")):
        compared, skipped = 0, []
        for k in range(1, t.K + 1):
            ref = op.solve(dtype, x0, fixed_iters=k, forced_steps=steps[:, :k], merit_log_len=k)
            mask, _ = t.conditioning(op, x0, x0_nudged, steps, k, dtype, ref, B)
            compared += int(mask.sum())
            skipped += [[k, int(b)] for b in range(B) if not mask[b]]
        out["%s:%s" % (scene, name)] = {"compared": compared, "of": B * t.K, "skipped": skipped}
        print(scene, name, compared, "/", B * t.K)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "forced_coverage.json"), "w"), indent=1)
