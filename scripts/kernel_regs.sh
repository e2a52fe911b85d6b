#!/bin/bash
# Registers / LDS / scratch of the device kernels of one (n, N, m_i) instantiation:  scripts/kernel_regs.sh 24 4 2 [pattern]
# (also leaves the device assembly in /tmp/ilqg_regs_<n>_<N>_<mu>/dev.s)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/ilqg_regs_$1_$2_$3
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I$ROOT/include -DILQG_PART_NX=$1 -DILQG_PART_NP=$2 -DILQG_PART_MU=$3 \
  -Wno-unused-function --cuda-device-only -S $ROOT/ilqgames_amd/csrc/ilqg_api.hip -o $OUT/dev.s
python3 - "$OUT/dev.s" "${4:-.}" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2], re.I)
for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
    blk = ".agpr_count:" + blk
    f = dict(re.findall(r"\.(agpr_count|name|vgpr_count|sgpr_count|vgpr_spill_count|private_segment_fixed_size|max_flat_workgroup_size):\s+(\S+)", blk))
    if "name" in f and pat.search(f["name"]):
        print("vgpr %4s agpr %4s sgpr %4s spill %4s scratch %5s wg %5s  %s" % (f.get("vgpr_count"), f.get("agpr_count"), f.get("sgpr_count"),
              f.get("vgpr_spill_count"), f.get("private_segment_fixed_size"), f.get("max_flat_workgroup_size"), f["name"][:150]))
PY
