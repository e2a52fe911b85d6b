"""Diagnostic: the batch as S slices on S streams, offset so that one slice's sweep overlaps another's trial pass."""
import argparse, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from ilqgames_amd import abi, examples, hip
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--streams", type=int, default=2)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dtype = abi.F64 if a.dtype == "f64" else abi.F32
spec = examples.CONFIGS["modified_three_player_intersection"]()
spec.params.initial_alpha_scaling = 0.1; spec.params.expected_decrease_fraction = 0.001; spec.params.max_backtracking_steps = 100
S = a.streams
Bs = a.batch // S
probs = [hip.Problem(spec, dtype) for _ in range(S)]
x0 = torch.as_tensor(examples.jittered_x0(spec, a.batch, seed=0), dtype=hip.torch_dtype(dtype), device="cuda")
bufs = [p.alloc_solve_buffers(Bs) for p in probs]
streams = [torch.cuda.Stream() for _ in range(S)]
def run():
    for i in range(S):
        with torch.cuda.stream(streams[i]):
            probs[i].solve(x0[i * Bs:(i + 1) * Bs], bufs[i], fixed_iters=a.iters)
torch.cuda.synchronize(); run(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    for b in bufs:
        for k in ("xs", "us", "P", "alpha"): b[k].zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in streams: s.wait_event(e0)
    run()
    for s in streams: torch.cuda.current_stream().wait_stream(s)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort(); ms = ts[2]
print("batch %d %s on %d streams: %.3f ms per solve, %.3f ms per iteration, %.3f M it/s" % (a.batch, a.dtype, S, ms, ms / a.iters, a.batch * a.iters / ms / 1e3))
