timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forced.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_receding.py -x -q -m gpu -k "fp32_device_against or config5_receding" -s 2>&1 | grep -E "config 5 fp32|passed|failed|Error|assert" | head
for dt in f64 f32; do timeout 120 python scripts/exp_modes.py --batch 1024 --dtype $dt --iters 10 --reps 5 2>&1 | tail -1; timeout 120 python scripts/exp_modes.py --batch 8192 --dtype $dt --iters 6 --reps 3 2>&1 | tail -1; done
