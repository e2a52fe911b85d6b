timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "test_ilq_solve_matches_oracle_fp64" 2>&1 | tail -6
