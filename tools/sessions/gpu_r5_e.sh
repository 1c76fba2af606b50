#!/bin/bash
# round 5, session E: the f32 matrix pipe under realistic operands (random mantissas), and the tile kernels' own durations.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
tools/microbench/mfma_f32_peak > $O/r5e_mfma_f32_peak.txt 2>&1; echo "mfma rc $?"; cat $O/r5e_mfma_f32_peak.txt
for kern in 2 4; do
  rm -rf /tmp/prof_x$kern
  (cd /tmp && VSS_EXACT_KERNEL=$kern timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_x$kern -o x -- python $GRAFT_REPO_ROOT/tools/gpu_exact_probe.py 2000000 > /tmp/prof_x$kern.log 2>&1)
  echo "rocprof kernel=$kern rc $?"; grep -v "rocprofv3\|amdgpu.ids" /tmp/prof_x$kern.log | tail -n 6
  find /tmp/prof_x$kern -name "*.csv" | head
  f=$(find /tmp/prof_x$kern -name "*kernel_stats.csv" | head -n 1)
  [ -n "$f" ] && head -n 8 "$f" | cut -c1-200 > $O/r5e_exact_kernel${kern}_stats.csv && cat $O/r5e_exact_kernel${kern}_stats.csv
done
