#!/bin/bash
# round 5, session F: where do the score tile's cycles go?  Kernel-only durations (rocprofv3 kernel trace) of k_exact_scores_v2 and
# _v3 on 1024 x 2M x 768, and of v3 with parts switched off (VSS_EXACT_PROBE: 1 no score stores / survivors, 2 no global loads
# after the prologue, 4 no barrier, 8 no stagger; answers wrong by design).
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
ROWS=2000000
FLOPS=$(python -c "print(4 * 2.0 * 1024 * $ROWS * 768)")   # gpu_exact_probe.py: one warm-up batch + three timed ones
: > $O/r5f_exact_tile_ablation.txt
for cfg in "2 0" "4 0" "4 8" "4 1" "4 2" "4 3" "4 7" "2 3" "2 7"; do
  set -- $cfg
  rm -rf /tmp/prof_x
  (cd /tmp && VSS_EXACT_KERNEL=$1 VSS_EXACT_PROBE=$2 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_x -o x -- python $GRAFT_REPO_ROOT/tools/gpu_exact_probe.py $ROWS > /tmp/prof_x.log 2>&1)
  echo "kernel $1 probe $2: $(grep 'exact top' /tmp/prof_x.log)" >> $O/r5f_exact_tile_ablation.txt
  python tools/rocprof_kernel_table.py /tmp/prof_x k_exact $FLOPS >> $O/r5f_exact_tile_ablation.txt 2>&1
done
cat $O/r5f_exact_tile_ablation.txt
