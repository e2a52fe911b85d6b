"""Wall-clock timeline of one round of the fused solve (diagnostic; needs a -DILQG_TIMELINE=1 build:
   python scripts/devbuild.py --tag tl --dims 14,3,2 -- -DILQG_TIMELINE=1
   ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_tl.so python scripts/timeline.py [--batch 1024 --dtype f64])."""
import argparse, ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from ilqgames_amd import abi, examples, hip
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--config", default="modified_three_player_intersection")
a = ap.parse_args()
dtype = abi.F64 if a.dtype == "f64" else abi.F32
spec = examples.CONFIGS[a.config]()
spec.params.initial_alpha_scaling = 0.1; spec.params.expected_decrease_fraction = 0.001; spec.params.max_backtracking_steps = 100
B = a.batch
prob = hip.Problem(spec, dtype)
x0 = torch.as_tensor(examples.jittered_x0(spec, B, seed=0), dtype=hip.torch_dtype(dtype), device="cuda")
bufs = prob.alloc_solve_buffers(B)
prob.solve(x0, bufs, fixed_iters=4); torch.cuda.synchronize()
prof = torch.zeros((B, 96), dtype=torch.int64, device="cuda")
hip.lib().ilqg_debug_set_profile_buffer(ctypes.c_void_p(prof.data_ptr()))
for k in ("xs", "us", "P", "alpha"): bufs[k].zero_()
prob.solve(x0, bufs, fixed_iters=4); torch.cuda.synchronize()
hip.lib().ilqg_debug_set_profile_buffer(None)
tl = prof[:, 32:96].double().cpu().numpy() / 100.0  # microseconds
def d(i, j):
    v = tl[:, j] - tl[:, i]
    ok = (tl[:, i] > 0) & (tl[:, j] > 0)
    return "%7.1f (p10 %7.1f p90 %7.1f)" % (np.mean(v[ok]), np.percentile(v[ok], 10), np.percentile(v[ok], 90)) if ok.any() else "n/a"
t0 = tl[:, 0].min()
print("last trial launch, us relative to each instance's own entry (mean over %d instances; p10 / p90):" % B)
print("  entry spread over the batch: %.1f us" % (tl[:, 0].max() - t0))
for name, i, j in (("entry -> pass start", 0, 1), ("rollout (wave 0)", 1, 2), ("  steps 0-32", 20, 21), ("  steps 32-64", 21, 22), ("  steps 64-96", 22, 23),
                   ("entry -> forward pass done (row wave)", 0, 3), ("chunk 0 wait (fwd done -> start)", 3, 4), ("chunk 0", 4, 5),
                   ("chunk 0 end -> chunk 1 start", 5, 6), ("chunk 1", 6, 7), ("rollout end -> chunk 1 start", 2, 6), ("chunk 1 end -> reductions", 7, 12),
                   ("reductions + decision", 12, 13), ("  merit + cost reduction", 12, 14), ("  decision, cost commit", 14, 15),
                   ("  state store, end", 15, 13), ("whole pass (entry -> end)", 0, 13)):
    print("  %-44s %s" % (name, d(i, j)))
print("last chunk of the row wave:")
for name, i, j in (("chunk start -> (x, u) staged", 6, 40), ("persistent slot init", 40, 41), ("Jacobian pass: ops", 41, 42), ("Jacobian pass: write-out", 42, 43),
                   ("player 0: ops", 43, 44), ("player 0: write-out", 44, 45), ("player 1: ops", 45, 46), ("player 1: write-out", 46, 47),
                   ("player 2: ops", 47, 48), ("player 2: write-out", 48, 49)):
    print("  %-44s %s" % (name, d(i, j)))
print("last sweep launch:")
for name, i, j in (("entry -> loop", 16, 17), ("loop (T-1 steps)", 17, 18), ("loop end -> exit", 18, 19), ("whole", 16, 19)):
    print("  %-44s %s" % (name, d(i, j)))
print("  kernel span over the batch: trial %.1f us, sweep %.1f us" % (tl[:, 13].max() - tl[:, 0].min(), tl[:, 19].max() - tl[:, 16].min()))

# SIMD placement of the sweep's waves (HW_ID) against the instance's loop time
hw = prof[:, 32 + 24:32 + 27].cpu().numpy()
loop = tl[:, 18] - tl[:, 17]
cu_of = {}
for b in range(B):
    h = int(hw[b, 0]); xcc = (h >> 32) & 15; h &= 0xffffffff
    key = (xcc, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15)
    cu_of.setdefault(key, []).append(b)
bal, unb = [], []
crowd = {2: [], 3: [], 4: [], 5: []}
for key, bs in cu_of.items():
    cnt = [0, 0, 0, 0]
    for b in bs:
        for w in range(3):
            cnt[(int(hw[b, w]) >> 4) & 3] += 1
    (bal if max(cnt) == min(cnt) else unb).extend(loop[bs])
    for b in bs:  # the most crowded SIMD this instance has a wave on
        m = max(cnt[(int(hw[b, w]) >> 4) & 3] for w in range(3))
        crowd.setdefault(m, []).append(loop[b])
print("sweep loop time by CU balance: %d instances on balanced CUs mean %.1f us; %d on unbalanced CUs mean %.1f us" %
      (len(bal), np.mean(bal) if bal else 0, len(unb), np.mean(unb) if unb else 0))
for m in sorted(crowd):
    if crowd[m]: print("  instances whose most crowded SIMD holds %d waves: %d, mean loop %.1f us" % (m, len(crowd[m]), np.mean(crowd[m])))

by_xcc = {}
within = []
cu_means = []
for key, bs in cu_of.items():
    by_xcc.setdefault(key[0], []).extend(loop[bs])
    within.append(np.max(loop[bs]) - np.min(loop[bs]))
    cu_means.append(np.mean(loop[bs]))
print("  per XCD mean loop us:", " ".join("%d:%.0f" % (x, np.mean(v)) for x, v in sorted(by_xcc.items())))
print("  within-CU spread (max - min) mean %.1f us; CU means: min %.1f p50 %.1f max %.1f" %
      (np.mean(within), np.min(cu_means), np.median(cu_means), np.max(cu_means)))
# does the time depend on the block index / entry time?
order = np.argsort(tl[:, 16])
q = B // 4
print("  loop us by launch order quartile:", " ".join("%.0f" % np.mean(loop[order[i * q:(i + 1) * q]]) for i in range(4)))
print("  loop us by block index quartile:", " ".join("%.0f" % np.mean(loop[i * q:(i + 1) * q]) for i in range(4)))
os.makedirs(os.path.join(R, "gpurun_out", "tl"), exist_ok=True); np.save(os.path.join(R, "gpurun_out", "tl", "tl_raw.npy"), prof.cpu().numpy())
