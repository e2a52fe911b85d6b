ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_b3.so timeout 200 python scripts/quick_parity.py 2>&1 | grep -v amdgpu | tail -9
for lib in b3 c b3 c; do
  ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_$lib.so timeout 120 python scripts/exp_modes.py --batch 1024 --dtype f64 --iters 10 --reps 5 2>&1 | tail -1
  ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_$lib.so timeout 120 python scripts/exp_modes.py --batch 1024 --dtype f32 --iters 10 --reps 5 2>&1 | tail -1
done
