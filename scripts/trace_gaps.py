"""Idle time between consecutive kernel dispatches of a rocprofv3 --kernel-trace database, by (previous -> next) kernel:
   python scripts/trace_gaps.py gpurun_out/prof_TAG/trace/run_results.db [last_n_dispatches]"""
import collections
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
rows = cur.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                   "on d.kernel_id = s.id order by d.start").fetchall()
if len(sys.argv) > 2:
    rows = rows[-int(sys.argv[2]):]
short = lambda n: re.sub(r"<.*", "", n.replace("void (anonymous namespace)::", "").replace("void ilqg::", ""))[:28]  # noqa: E731
gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    g = b[0] - a[1]
    if g < 500000:
        gap[(short(a[2]), short(b[2]))].append(g)
busy = sum(r[1] - r[0] for r in rows)
span = rows[-1][1] - rows[0][0]
print("dispatches %d, span %.2f ms, busy %.2f ms, idle %.2f ms" % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6))
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:16]:
    v.sort()
    print("%-28s -> %-28s n %5d  total %8.1f us  median %6.1f us" % (k[0], k[1], len(v), sum(v) / 1e3, v[len(v) // 2] / 1e3))
