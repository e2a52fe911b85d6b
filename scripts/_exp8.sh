timeout 600 python bench.py > gpurun_out/bench_default2.txt 2>gpurun_out/bench_default2.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_default2.txt') if l.startswith('{')][-1])
print("value %.4g ms_per_step %.4f frac %.4f reps %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["repeats"]["ms_per_step_all"]))
print("latency", d.get("latency")); print("cpu", {k:d["cpu_baseline"][k] for k in ("value","value_all_cores","cores_all","sample_all_cores")})
PY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forced.py -x -q -m gpu 2>&1 | tail -2
