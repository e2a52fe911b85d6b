"""Top kernels of a `rocprofv3 --kernel-trace --stats -d <dir> -o bench` run (the rocpd database it leaves)."""
import glob
import os
import sqlite3
import sys

for db in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)):
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print("##", db)
    for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
        print("| %s | %d | %.1f us total | %.1f us avg | %.2f %% |" % (r[0][:100], r[1], r[2], r[3], r[4]))
    con.close()
