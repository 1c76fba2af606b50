#!/bin/bash
# round 5, session G: ablation of k_exact_scores_v3 in the PLAIN mode (VSS_EXACT_FILTER=0: one launch per 32768-row chunk, scores
# stored, no overflow logic that wrong answers could trip): kernel-only TFLOP/s with parts switched off (VSS_EXACT_PROBE bits:
# 1 no score stores, 2 no global loads after the prologue, 4 no barrier, 16 no LDS writes after the prologue, 32 no epilogue).
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
ROWS=2000000
FLOPS=$(python -c "print(4 * 2.0 * 1024 * $ROWS * 768)")
: > $O/r5g_exact_tile_ablation_plain_mode.txt
for cfg in "2 0" "4 0" "4 1" "4 33" "4 2" "4 35" "4 51" "4 55" "4 48" "4 16"; do
  set -- $cfg
  rm -rf /tmp/prof_x
  (cd /tmp && VSS_EXACT_FILTER=0 VSS_EXACT_KERNEL=$1 VSS_EXACT_PROBE=$2 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_x -o x -- python $GRAFT_REPO_ROOT/tools/gpu_exact_probe.py $ROWS > /tmp/prof_x.log 2>&1)
  echo "kernel $1 probe $2: $(python tools/rocprof_kernel_table.py /tmp/prof_x k_exact_scores $FLOPS)" >> $O/r5g_exact_tile_ablation_plain_mode.txt
done
cat $O/r5g_exact_tile_ablation_plain_mode.txt
