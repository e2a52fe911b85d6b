ILQG_HIP_LIB=$PWD/ilqgames_amd/libilqg_hip_prof.so python scripts/stage_bench.py 2>&1 | grep -E "trial wave 1"
