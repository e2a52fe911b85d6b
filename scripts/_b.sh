timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forced.py -q -x 2>&1 | tail -4
timeout 600 python scripts/stage_bench.py 2>&1 | head -5
for i in 1 2; do timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency 2>&1 | tail -1 | cut -c80-200; done
timeout 600 python bench.py --steps 10 --warmup 2 --dtype f32 --no-cpu-baseline --no-latency 2>&1 | tail -1 | cut -c80-200
timeout 600 python bench.py --config roundabout_merging_T150 --batch 4096 --steps 4 --warmup 1 --no-cpu-baseline --no-latency 2>&1 | tail -1 | cut -c80-200
