"""Turns the rocprofv3 databases of scripts/profile.sh into the text summaries kept under profiles/."""
import json
import os
import sqlite3
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
out = []


def q(db, sql):
    con = sqlite3.connect(os.path.join(src, db))
    cur = con.cursor()
    rows = cur.execute(sql).fetchall()
    cols = [d[0] for d in cur.description]
    con.close()
    return cols, rows


out.append("# rocprofv3 --kernel-trace --stats  (python bench.py %s)\n" % os.environ.get("BENCH_ARGS", "--steps 10 --warmup 2 --no-cpu-baseline --no-latency"))
cols, rows = q("trace/bench_results.db", "select name, total_calls, total_duration, average, percentage from top_kernels")
out.append("| kernel | calls | total us | average us | % |\n|---|---|---|---|---|")
for r in rows:
    out.append("| %s | %d | %.1f | %.1f | %.2f |" % (r[0][:110], r[1], r[2], r[3], r[4]))
# the batch's dispatches only (grid = batch x workgroup size); bench.py's single-instance latency figure adds
# thousands of one-workgroup dispatches of the same kernels, which are summarised separately
cols, rows = q("trace/bench_results.db",
               "select name, duration, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count "
               "from kernels where name like '%ilq%' order by start")
big = [r for r in rows if r[2] > 4 * r[3]]
small = [r for r in rows if r[2] <= 4 * r[3]]
out.append("\n## dispatches of the dominant kernels (batch launches; the last %d shown of %d)\n\n| kernel | duration ns | grid | wg | LDS B | scratch B | VGPR | AGPR | SGPR |\n|---|---|---|---|---|---|---|---|---|" % (min(len(big), 30), len(big)))
for r in big[-30:]:
    out.append("| %s | %d | %d | %d | %d | %d | %d | %d | %d |" % ((r[0][:80],) + tuple(r[1:])))
import statistics
out.append("\n## per-kernel averages over the batch launches of the timed solve\n\n| kernel | launches | mean us | median us |\n|---|---|---|---|")
names = sorted(set(r[0] for r in big))
for nm in names:
    d = [r[1] for r in big if r[0] == nm]
    out.append("| %s | %d | %.1f | %.1f |" % (nm[:90], len(d), statistics.mean(d) / 1e3, statistics.median(d) / 1e3))
if small:
    out.append("\nSingle-instance dispatches (bench.py's `latency` figure, one workgroup each): %d, mean %.1f us." %
               (len(small), statistics.mean(r[1] for r in small) / 1e3))
for name, db in (("FETCH_SIZE", "pmc_fetch/bench_results.db"), ("WRITE_SIZE", "pmc_write/bench_results.db")):
    if not os.path.exists(os.path.join(src, db)):
        continue
    cols, rows = q(db, "select kernel_name, counter_name, value, duration from counters_collection "
                       "where kernel_name like '%ilq%' order by start")
    out.append("\n## rocprofv3 --pmc %s (own pass, no trace domains): medians per kernel over the batch launches\n\n| kernel | counter | launches | median value (KB, raw) | median duration ns |\n|---|---|---|---|---|" % name)
    for nm in sorted(set(r[0] for r in rows)):
        sel = [r for r in rows if r[0] == nm]
        top = max(r[2] for r in sel)
        sel = [r for r in sel if r[2] > 0.2 * top]  # drops the one-workgroup dispatches of the latency figure
        out.append("| %s | %s | %d | %.1f | %d |" % (nm[:80], sel[0][1], len(sel), statistics.median(r[2] for r in sel),
                                                    statistics.median(r[3] for r in sel)))
for f in ("bench_plain.log",):
    p = os.path.join(src, f)
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{"):
                out.append("\n## bench.py line of the same command (un-profiled run)\n\n```json\n%s```" % json.dumps(json.loads(line), indent=1)[:4000])
os.makedirs("profiles", exist_ok=True)
path = "profiles/%s.md" % tag
open(path, "w").write("\n".join(out) + "\n")
print(open(path).read())
