#!/bin/bash
# round 3, GPU session Y: rocprofv3 kernel trace of the configs[1] line (which kernel answers the one-query probe, how long)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 100 rocprofv3 --kernel-trace --stats -d /tmp/c2prof -o c2 -- python3 $R/bench.py --config c2 --steps 2000 --no-cpu-baseline > $R/gpurun_out/r3y_c2_under_rocprof.json 2> /tmp/c2prof.err; echo "rc $?"
cd $R && python - <<'PY'
import csv, sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/c2prof/**/c2_results.db", recursive=True)[0])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
with open("gpurun_out/r3y_c2_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, mn, mx in rows[:14]:
        w.writerow([name[:110], calls, tot, "%.1f" % avg, "%.4f" % (100.0 * tot / total), mn, mx])
for r in rows[:6]:
    print(r[0][:90], r[1], "avg %.1f us" % (r[3] / 1e3))
PY
