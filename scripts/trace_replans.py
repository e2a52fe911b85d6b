"""Per-replan kernel breakdown of a receding-horizon run traced with scripts/trace_cmd.sh (rocprofv3 --kernel-trace):
   python scripts/trace_replans.py gpurun_out/prof_TAG/trace/run_results.db
The dispatches between the second plan_integrate launch and the last are the warm-started replans (two integrations each)."""
import collections
import re
import sqlite3
import sys

import numpy as np

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                   "on d.kernel_id = s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if "plan_integrate" in r[2]]
seg = rows[idx[1]:idx[-1]]
n = (len(idx) - 2) / 2.0
agg = collections.defaultdict(list)
for s, e, k in seg:
    k = re.sub(r"\(.*", "", k.replace("void (anonymous namespace)::", "").replace("void ilqg::", ""))[:48]
    agg[k].append((e - s) / 1e3)
span = (seg[-1][1] - seg[0][0]) / 1e6
busy = sum(sum(v) for v in agg.values()) / 1e3
print("replans %.1f: span %.1f ms, kernels %.1f ms, %.0f launches per replan" % (n, span / n, busy / n, len(seg) / n))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v = np.array(v)
    print("%-50s n %6.1f  %6.2f ms  median %6.1f  p90 %6.1f us" % (k, len(v) / n, v.sum() / 1e3 / n, np.median(v), np.percentile(v, 90)))
