ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_c.so timeout 200 python scripts/quick_parity.py 2>&1 | grep -v amdgpu | tail -10
for c in 1 0; do
  ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_c.so timeout 120 python scripts/exp_modes.py --batch 1024 --dtype f64 --iters 10 --reps 5 --compact $c 2>&1 | tail -1
  ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_c.so timeout 120 python scripts/exp_modes.py --batch 1024 --dtype f32 --iters 10 --reps 5 --compact $c 2>&1 | tail -1
done
ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_c.so timeout 120 python scripts/exp_modes.py --batch 8192 --dtype f32 --iters 6 --reps 3 2>&1 | tail -1
ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_c.so timeout 120 python scripts/exp_modes.py --batch 8192 --dtype f64 --iters 6 --reps 3 2>&1 | tail -1
ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_tl.so timeout 100 python scripts/timeline.py 2>&1 | grep -v amdgpu | head -36
