"""Experiment / profile builds of libilqg_hip.so next to the product library (diagnostic).

  python scripts/devbuild.py --tag prof --dims 14,3,2 -- -DILQG_PROFILE=1
builds ilqgames_amd/libilqg_hip_prof.so holding only the listed instantiations; select it at run time with
ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_prof.so (ilqgames_amd/hip.py)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tag", required=True)
ap.add_argument("--dims", default="14,3,2", help="semicolon-separated n,N,mu triples")
ap.add_argument("flags", nargs="*")
a = ap.parse_args()
dims = [tuple(int(v) for v in d.split(",")) for d in a.dims.split(";")]
out = os.path.join(ROOT, "ilqgames_amd", "libilqg_hip_%s.so" % a.tag)
ge.build_hip_library(out=out, only_dims=dims, extra_flags=a.flags, objdir_name="_obj_" + a.tag)
print(out)
