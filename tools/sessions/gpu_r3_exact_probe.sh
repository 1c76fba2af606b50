#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03_exact
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for kern in ${KERNELS:-2 3}; do
VSS_EXACT_KERNEL=$kern timeout 300 python -m pytest $R/tests -q -m gpu -x -k "exact or golden or readme" -p no:cacheprovider 2>&1 | tail -n 1
for p in ${PROBES:-0 1 2 7}; do
  VSS_EXACT_KERNEL=$kern VSS_EXACT_PROBE=$p timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_p$p -o exact -- python $R/tools/gpu_exact_probe.py 1000000 > $O/exact_k${kern}_p$p.txt 2> $O/exact_k${kern}_p$p.err
  python - $O/kt_p$p $p $kern <<'PY'
import sqlite3, sys
d = sqlite3.connect(sys.argv[1] + "/exact_results.db")
full = d.execute("select avg(end-start), count(*) from kernels where name like '%k_exact_scores%' and (end-start) > 0.9 * (select max(end-start) from kernels where name like '%k_exact_scores%')").fetchone()
print("kernel", sys.argv[3], "probe", sys.argv[2], "full chunk avg %.1f us (%d launches) -> %.1f TFLOP/s = %.3f of 157.3" % (full[0] / 1e3, full[1], 2.0 * 1024 * 32768 * 768 / full[0] / 1e3, 2.0 * 1024 * 32768 * 768 / full[0] / 1e3 / 157.3))
PY
  rm -rf $O/kt_p$p
done
done
