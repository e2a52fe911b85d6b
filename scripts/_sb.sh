for mode in new old; do
  if [ $mode = old ]; then export ILQG_OLD_ROWS=1; else unset ILQG_OLD_ROWS; fi
  for dt in f64 f32; do echo "== $mode $dt"; python scripts/stage_bench.py --dtype $dt 2>&1 | grep -E "linearize|quadratize|totalcosts"; done
done
