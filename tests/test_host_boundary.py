"""Host-side checks that need no GPU: the C header and the ctypes mirrors agree, the built
library exports every symbol include/ilqg.h declares, the product path refuses to run without a
device (no CPU fallback), and the N>1 sharding + gather logic works over gloo."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_declares_what_the_loader_binds():
    from ilqgames_amd import hip
    hdr = open(os.path.join(ROOT, "include", "ilqg.h")).read()
    declared = set(re.findall(r"\b(ilqg_[a-z_0-9]+)\s*\(", hdr))
    assert set(hip.EXPORTS) <= declared
    assert declared - set(hip.EXPORTS) == set(), "header declares symbols the loader does not know"


def test_library_exports_every_declared_symbol():
    from ilqgames_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    # nm instead of dlopen: loading needs libamdhip64, and this must not initialise HIP on a CPU box
    out = subprocess.check_output(["nm", "-D", "--defined-only", hip.LIB_PATH], text=True)
    syms = set(line.split()[-1] for line in out.splitlines() if line.strip())
    missing = [s for s in hip.EXPORTS if s not in syms]
    assert not missing, missing


def test_library_has_no_dangling_internal_symbol():
    """Every undefined symbol of libilqg_hip.so must come from a library it links (HIP runtime, libstdc++, libc, libm,
    libgcc).  An undefined symbol of the library's OWN namespaces is a call through a null pointer at run time: the
    hidden-visibility units resolve it to address 0 at link time without a word (this happened with an `extern
    thread_local` of a constant-initialised class type, whose TLS init function no unit emits)."""
    from ilqgames_amd import hip
    out = subprocess.check_output(["nm", "-C", "--undefined-only", hip.LIB_PATH], text=True)
    own = [line.split(None, 1)[-1].strip() for line in out.splitlines() if "ilqg" in line]
    assert not own, own


def test_struct_layouts_match_the_c_header():
    """Compile a tiny C program against include/ilqg.h and compare sizeof/offsetof with ctypes."""
    from ilqgames_amd import abi
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "ilqg.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(ilqg_dims), sizeof(ilqg_subsystem), sizeof(ilqg_cost_term),
         sizeof(ilqg_player_cost), sizeof(ilqg_solver_params), sizeof(ilqg_problem_desc), sizeof(ilqg_pair));
  printf("%zu %zu %zu %zu\n", offsetof(ilqg_problem_desc, terms), offsetof(ilqg_problem_desc, dt),
         offsetof(ilqg_problem_desc, params), offsetof(ilqg_cost_term, constraint_slot));
  return 0;
}'''
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        a, b = subprocess.check_output([exe], text=True).strip().split("\n")
    sizes = [int(v) for v in a.split()]
    offs = [int(v) for v in b.split()]
    assert sizes == [C.sizeof(abi.Dims), C.sizeof(abi.Subsystem), C.sizeof(abi.CostTerm), C.sizeof(abi.PlayerCost),
                     C.sizeof(abi.SolverParams), C.sizeof(abi.ProblemDesc), C.sizeof(abi.Pair)]
    assert offs == [abi.ProblemDesc.terms.offset, abi.ProblemDesc.dt.offset, abi.ProblemDesc.params.offset,
                    abi.CostTerm.constraint_slot.offset]


def test_no_cpu_fallback():
    """Without a GPU the product entry points fail loudly (ILQG_ERR_NO_DEVICE), they never compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ilqgames_amd import abi, examples, hip
    try:
        lib = hip.lib()
    except OSError as e:  # libamdhip64 not loadable on this box: equally loud
        assert "hip" in str(e).lower() or "cannot open" in str(e).lower()
        return
    spec = examples.modified_three_player_intersection()
    with pytest.raises(hip.IlqgError) as e:
        hip.Problem(spec, abi.F64)
    assert e.value.status in (abi.ERR_NO_DEVICE, abi.ERR_HIP)


def test_product_path_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under ilqgames_amd/ may reference it."""
    pkg = os.path.join(ROOT, "ilqgames_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower().replace("test infrastructure", ""), os.path.join(dirpath, f)


_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from ilqgames_amd import sharding
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
B_total = 10
lo, hi = sharding.instance_range(B_total, rank, 2)
assert (lo, hi) == ((0, 5) if rank == 0 else (5, 10))
# every rank "solves" its shard: the strategy of instance b is filled with b
local = torch.stack([torch.full((7,), float(b)) for b in range(lo, hi)])
full = sharding.gather_to_root(local, B_total, 2)
if rank == 0:
    assert full.shape == (10, 7)
    assert torch.equal(full[:, 0], torch.arange(10, dtype=full.dtype))
else:
    assert full is None
# uneven split
lo, hi = sharding.instance_range(7, rank, 2)
local = torch.arange(lo, hi, dtype=torch.float64).reshape(-1, 1)
full = sharding.gather_to_root(local, 7, 2)
if rank == 0:
    assert torch.equal(full[:, 0], torch.arange(7, dtype=torch.float64))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_sharding_and_gather_over_gloo(tmp_path):
    """N>1 path on CPU: contiguous instance blocks per rank, gather of per-instance results to rank 0
    (the only exchange step; RCCL on the GPU box, gloo here)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "w.py"
    script.write_text(_WORKER % dict(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_bench_two_rank_path_over_gloo():
    """The code `bench.py --gpus N` executes for N > 1 — contiguous shards of the global batch, the gather of the
    strategies to rank 0 through ilqgames_amd/sharding.py, max-over-ranks timing, one JSON line from rank 0 — launched
    exactly as the driver launches it (torch.distributed.run, one process per rank), with the device solve replaced
    by the CPU stand-in of `--backend stub` and gloo in place of RCCL.  The stand-in's strategy of global instance b
    is b, and bench.py itself asserts on rank 0 that the gathered rows are 0 .. world * batch - 1 in order."""
    import json
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "stub", "--steps", "2",
           "--warmup", "1", "--batch", "5"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["value"] > 0 and abs(out["value"] * out["ms_per_step"] * 1e-3 * out["steps"] - 2 * 5 * 2) < 1e-6


def test_bench_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the form of the driver's N = 1 command with another N)
    must run TWO ranks, not one rank that prints n_gpus = 1: bench.py re-executes itself under torch.distributed.run.
    A launcher whose WORLD_SIZE disagrees with --gpus is an error."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "stub", "--steps", "2",
           "--warmup", "1", "--batch", "3"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 * out["steps"] - 2 * 3 * 2) < 1e-6
    bad = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=bad)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_fast_trig_of_the_rollout_is_within_its_documented_error(tmp_path):
    """csrc/ilqg_trig.hpp is __host__ __device__: tests/host/trig_check.cpp compiles it for the host (plain g++ against
    the HIP headers) and compares sine / cosine / tangent with the C library in long double over the whole fast-path
    range, dense around the multiples of pi/2 — <= 2 ulp (tangent 3) or the program exits non-zero — and, beyond that
    range, the large-argument reduction (trig_reduce_large) over every binade up to the largest finite double / float, next
    to multiples of pi/2, at the classical worst case of the reduction, and on +-inf / NaN."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "trig_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(root, "tests", "host", "trig_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "double:" in out.stdout and "float:" in out.stdout and "double beyond" in out.stdout and "float beyond" in out.stdout
