"""Experiment: the headline batch as two half batches on two HIP streams, the second a phase behind the first, so that
one half's sweep (instruction-issue-bound) runs beside the other half's trial pass (one latency chain per instance).
Each half is its own ilqg_problem / workspace / stream, driven by its own host thread.
usage (GPU box): python scripts/ab_two_streams.py [--batch 1024] [--steps 20] [--dtype f64] [--parts 2]"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import bench
    from ilqgames_amd import abi, examples, hip
    spec, _ = bench._bench_spec(examples, bench.HEADLINE_CONFIG, "auto")
    dtype = abi.F64 if args.dtype == "f64" else abi.F32
    B = args.batch
    x0 = torch.as_tensor(examples.jittered_x0(spec, B, seed=0), dtype=hip.torch_dtype(dtype), device="cuda")

    def timed(parts, delays_us):
        """`parts` contiguous pieces of the batch, piece q on its own stream, starting delays_us[q] after the others."""
        probs = [hip.Problem(spec, dtype) for _ in range(parts)]
        cuts = [B * q // parts for q in range(parts + 1)]
        x0s = [x0[cuts[q]:cuts[q + 1]].contiguous() for q in range(parts)]
        bufs = [probs[q].alloc_solve_buffers(cuts[q + 1] - cuts[q]) for q in range(parts)]
        streams = [torch.cuda.Stream() for _ in range(parts)]
        clock_hz = 2.4e9

        def run(q, steps, start, stop):
            with torch.cuda.stream(streams[q]):
                streams[q].wait_event(start)
                if delays_us[q] > 0:
                    torch.cuda._sleep(int(delays_us[q] * 1e-6 * clock_hz))
                probs[q].solve(x0s[q], bufs[q], fixed_iters=steps)
                stop[q].record(streams[q])

        out = []
        for rep in range(args.reps + 1):
            for b in bufs:
                for k in ("xs", "us", "P", "alpha"):
                    b[k].zero_()
            torch.cuda.synchronize()
            start = torch.cuda.Event(enable_timing=True)
            stop = [torch.cuda.Event(enable_timing=True) for _ in range(parts)]
            start.record()
            th = [threading.Thread(target=run, args=(q, args.steps, start, stop)) for q in range(parts)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            if rep:
                out.append(max(start.elapsed_time(s) for s in stop) - max(delays_us) * 0.0)
        iters = sum(int(b["iters"].sum().item()) for b in bufs)
        ms = sorted(out)[len(out) // 2]
        return ms, iters, bufs

    base_ms, base_iters, base_bufs = timed(1, [0])
    print("one stream: %.3f ms per iteration of the batch, %.0f it/s" % (base_ms / args.steps, base_iters / base_ms * 1e3))
    ref = {k: base_bufs[0][k].clone() for k in ("xs", "P", "alpha")}
    for delay in (0, 100, 200, 250, 300, 400):
        delays = [delay * q for q in range(args.parts)]
        ms, iters, bufs = timed(args.parts, delays)
        # the delayed piece's sleep is inside its own time; a long run amortises it
        same = all(torch.equal(torch.cat([b[k] for b in bufs]), ref[k]) for k in ref)
        print("%d streams, phase %3d us: %.3f ms per iteration of the batch, %.0f it/s, identical results: %s" % (
            args.parts, delay, ms / args.steps, iters / ms * 1e3, same))


if __name__ == "__main__":
    t0 = time.time()
    main()
    print("wall %.1f s" % (time.time() - t0))
