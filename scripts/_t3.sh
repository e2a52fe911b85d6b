export ILQG_HIP_LIB=$PWD/ilqgames_amd/libilqg_hip_a.so
python bench.py --no-cpu-baseline --no-latency 2>&1 | tail -1 | cut -c1-130
python bench.py --no-cpu-baseline --no-latency --dtype f32 2>&1 | tail -1 | cut -c1-130
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "(ilq_solve_matches_oracle_fp64 and modified_three) or (stage_kernels and modified_three) or (lq_feedback and 14)" 2>&1 | tail -3
ILQG_HIP_LIB=$PWD/ilqgames_amd/libilqg_hip_prof.so python scripts/stage_bench.py 2>&1 | grep -E "lq_feedback|per launch|wave [012] |trial wave 1|rollout"
