ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_c.so timeout 200 python scripts/quick_parity.py 2>&1 | grep -v amdgpu | tail -3
ILQG_HIP_LIB=ilqgames_amd/libilqg_hip_c.so timeout 200 python scripts/exp_latency.py 2>&1 | grep -v amdgpu | tail -5
