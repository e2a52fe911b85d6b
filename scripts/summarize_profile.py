"""Turns the rocprofv3 databases of scripts/profile.sh into the text summaries kept under profiles/."""
import json
import re
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
        sel = [r for r in sel if r[2] > 0.2 * top] or sel  # drops the one-workgroup dispatches of the latency figure
        out.append("| %s | %s | %d | %.1f | %d |" % (nm[:80], sel[0][1], len(sel), statistics.median(r[2] for r in sel),
                                                    statistics.median(r[3] for r in sel)))
# wave-level SQ counters (own pass): per kernel, medians over the batch launches
sqdb = "pmc_sq/bench_results.db"
if os.path.exists(os.path.join(src, sqdb)):
    cols, rows = q(sqdb, "select kernel_name, counter_name, value, duration from counters_collection "
                         "where kernel_name like '%ilq%' order by start")
    names = sorted(set(r[0] for r in rows))
    ctrs = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
            "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"]
    out.append("\n## rocprofv3 --pmc " + " ".join(ctrs) + " (own pass): medians per kernel over the batch launches; the "
               "WAIT / ACTIVE counters as a share of SQ_WAVE_CYCLES (all in quad-cycles)\n\n| kernel | waves | busy cycles | "
               "wave cycles | waiting (s_waitcnt / barrier) | issue-stalled | executing any | of which VALU | LDS |\n|---|---|---|---|---|---|---|---|---|")
    for nm in names:
        med = {}
        for c in ctrs:
            vals = [r[2] for r in rows if r[0] == nm and r[1] == c]
            if vals:
                top = max(vals)
                vals = [v for v in vals if v > 0.2 * top] or vals
                med[c] = statistics.median(vals)
        if "SQ_WAVE_CYCLES" not in med or med["SQ_WAVE_CYCLES"] <= 0:
            continue
        wc = med["SQ_WAVE_CYCLES"]
        sh = lambda c: "%.0f %%" % (100.0 * med.get(c, 0.0) / wc)  # noqa: E731
        out.append("| %s | %.0f | %.3g | %.3g | %s | %s | %s | %s | %s |" % (nm[:80], med.get("SQ_WAVES", 0), med.get("SQ_BUSY_CYCLES", 0), wc,
                   sh("SQ_WAIT_ANY"), sh("SQ_WAIT_INST_ANY"), sh("SQ_ACTIVE_INST_ANY"), sh("SQ_ACTIVE_INST_VALU"), sh("SQ_ACTIVE_INST_LDS")))
# HBM traffic per round for bench.py's roofline block: 2 x FETCH_SIZE (profiles/r03_counter_calibration.md: the counter
# reports half of the bytes read) + WRITE_SIZE, summed over the kernels of one round (sweep + trial kernels)
try:
    tr = {}
    for cname, db in (("FETCH_SIZE", "pmc_fetch/bench_results.db"), ("WRITE_SIZE", "pmc_write/bench_results.db")):
        cols, rows = q(db, "select kernel_name, counter_name, value, duration from counters_collection where kernel_name like '%ilq%' order by start")
        for nm in sorted(set(r[0] for r in rows)):
            if "exit" in nm or "probe" in nm:
                continue
            sel = [r for r in rows if r[0] == nm]
            top = max(r[2] for r in sel)
            sel = [r for r in sel if r[2] > 0.2 * top] or sel
            mm = re.search(r"(ilq_\w+?_kernel)", nm)
            short = mm.group(1) if mm else nm[:40]
            tr.setdefault(short, {})[cname] = statistics.median(r[2] for r in sel) * 1024.0
    total = sum(2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0) for v in tr.values())
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench  # csrc_sha16: which kernels these passes ran on (profiles/traffic.json entries carry it)
    out.append("\n## HBM traffic per round (2 x FETCH_SIZE + WRITE_SIZE over the round's kernels, profiles/r03_counter_calibration.md)\n\n```json\n%s\n```"
               % json.dumps({"bytes_per_round": total, "per_kernel_bytes": tr, "csrc_sha16": bench.csrc_sha16()}, indent=1))
except Exception as e:  # a pass that did not run leaves the section out
    out.append("\n(traffic per round not computed: %r)" % (e,))
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
