"""Per-kernel totals of a rocprofv3 --kernel-trace result directory (the rocpd SQLite database rocprofv3 7.x writes).
    python tools/rocprof_kernel_table.py <dir> [name filter] [flops for a TFLOP/s column]"""
import glob
import os
import sqlite3
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
flops = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
dbs = sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))
if not dbs:
    sys.exit("no .db under %s" % src)
db = sqlite3.connect(dbs[0])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                  "group by name order by 3 desc").fetchall()
for name, calls, tot, avg, mn, mx in rows:
    if flt in name:
        extra = "  %.1f TFLOP/s" % (flops / (tot * 1e-9) / 1e12) if flops else ""
        print("%-60s calls %5d  total %10.3f ms  avg %9.3f ms  min %9.3f  max %9.3f%s" % (name[:60], calls, tot / 1e6, avg / 1e6, mn / 1e6, mx / 1e6, extra))
