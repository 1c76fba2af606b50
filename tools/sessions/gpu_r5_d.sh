#!/bin/bash
# round 5, session D: k_exact_scores_v3 (persistent score tile) — the collected exact tests, A/B against v2 over wall clock, and
# the tile kernels' own durations from a rocprofv3 kernel trace.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
(time timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "exact") > $O/r5d_pytest_exact.txt 2>&1; echo "pytest rc $?"; tail -n 4 $O/r5d_pytest_exact.txt
timeout 600 python tools/gpu_exact_v3_probe.py 4000000 > $O/r5d_exact_v3_ab_4m768.txt 2>&1; echo "probe rc $?"; cat $O/r5d_exact_v3_ab_4m768.txt | grep -v amdgpu.ids
for kern in 2 4; do
  rm -rf /tmp/prof_x$kern
  (cd /tmp && VSS_EXACT_KERNEL=$kern timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x$kern -o x -- python $GRAFT_REPO_ROOT/tools/gpu_exact_probe.py 2000000 > /tmp/prof_x$kern.log 2>&1)
  echo "rocprof kernel=$kern rc $?"; tail -n 2 /tmp/prof_x$kern.log
  f=$(find /tmp/prof_x$kern -name "*kernel_stats.csv" | head -n 1)
  [ -n "$f" ] && head -n 8 "$f" | cut -c1-200 > $O/r5d_exact_kernel${kern}_stats.csv && cat $O/r5d_exact_kernel${kern}_stats.csv
done
