timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "costates or lq_" 2>&1 | tail -8
