timeout 600 python scripts/stage_bench.py 2>&1 | head -12
timeout 600 python bench.py --config three_player_collision_avoidance_reachability --batch 2048 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 | cut -c1-330
