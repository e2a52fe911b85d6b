timeout 1500 python -m pytest tests/test_gpu_receding.py -q -k "simulate_matches or config5" 2>&1 | tail -8
