"""Rebuilds profiles/traffic.json from the profile summaries of one round (profiles/<tag>_*.md, written by
scripts/summarize_profile.py): the "HBM traffic per round" block of each summary, keyed the way bench.py looks it up
(config:dtype:batch, read from the bench.py line the summary quotes) and stamped with the kernel-source hash of the
build the PMC passes ran on.

  python scripts/collect_traffic.py r04"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
out = {}
for path in sorted(glob.glob(os.path.join(ROOT, "profiles", tag + "_*.md"))):
    text = open(path).read()
    blocks = re.findall(r"```json\n(.*?)```", text, re.S)
    traffic = line = None
    for b in blocks:
        try:
            j = json.loads(b)
        except ValueError:
            continue
        if "bytes_per_round" in j:
            traffic = j
        elif "metric" in j:
            line = j
    if not traffic or not line or not traffic.get("bytes_per_round"):
        print("skipped (no traffic block or bench line):", os.path.basename(path))
        continue
    wl = line["config"]["workload"]
    m = re.match(r"(\S+) n=\d+ N=\d+ T=\d+ batch=(\d+)/GPU (f32|f64)", wl)
    if not m:
        print("skipped (workload not recognised):", os.path.basename(path), wl[:60])
        continue
    key = "%s:%s:%s" % (m.group(1), m.group(3), m.group(2))
    rel = os.path.relpath(path, ROOT)
    out[key] = {
        "source": rel + " (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, medians over the batch launches; "
                        "2 x FETCH_SIZE + WRITE_SIZE, profiles/r03_counter_calibration.md)",
        "collected": "round %s" % tag.lstrip("r0"),
        "csrc_sha16": traffic.get("csrc_sha16"),
        "per_kernel_bytes": traffic["per_kernel_bytes"],
        "bytes_per_round": traffic["bytes_per_round"],
    }
    print(key, "%.3g bytes / round" % traffic["bytes_per_round"], traffic.get("csrc_sha16"))
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
