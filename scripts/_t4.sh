for sp in 0 1; do
echo "== split=$sp"
ILQG_SPLIT_TRIAL=$sp python bench.py --no-cpu-baseline --no-latency --config roundabout_merging_T150 --batch 4096 --steps 4 --warmup 1 2>&1 | tail -1 | cut -c1-130
ILQG_SPLIT_TRIAL=$sp python bench.py --no-cpu-baseline --no-latency --batch 8192 --steps 10 2>&1 | tail -1 | cut -c1-130
ILQG_SPLIT_TRIAL=$sp python bench.py --no-cpu-baseline --no-latency --batch 1024 --steps 10 2>&1 | tail -1 | cut -c1-130
ILQG_SPLIT_TRIAL=$sp python bench.py --no-cpu-baseline --no-latency --batch 8192 --dtype f32 --steps 10 2>&1 | tail -1 | cut -c1-130
done
