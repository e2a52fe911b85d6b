for lib in w4 c w4 c; do
  ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_$lib.so timeout 120 python scripts/exp_modes.py --batch 8192 --dtype f32 --iters 6 --reps 3 2>&1 | tail -1
  ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_$lib.so timeout 120 python scripts/exp_modes.py --batch 1024 --dtype f32 --iters 10 --reps 5 2>&1 | tail -1
done
