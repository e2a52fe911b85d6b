set -x
timeout 900 python -m pytest tests -q -x -m gpu -k "open or roundabout or config4" 2>&1 | tail -5
timeout 600 python bench.py --config roundabout_merging_T150 --batch 4096 --steps 4 --warmup 1 --no-cpu-baseline --no-latency 2>&1 | tail -1 | cut -c1-330
timeout 600 python bench.py --config roundabout_merging_T150 --batch 4096 --dtype f32 --steps 4 --warmup 1 --no-cpu-baseline --no-latency 2>&1 | tail -1 | cut -c1-330
