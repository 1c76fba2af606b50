#!/bin/bash
# round 5, session H: k_exact_scores_v4 (persistent tile, operands by LDS-DMA, lean epilogue) — collected exact tests with it as
# the default, A/B over wall clock against v2 / v3, kernel-only durations, and its own ablation in the plain mode.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
(time timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "exact or smoke or recall") > $O/r5h_pytest_exact.txt 2>&1; echo "pytest rc $?"; tail -n 4 $O/r5h_pytest_exact.txt
timeout 600 python tools/gpu_exact_v3_probe.py 4000000 > $O/r5h_exact_v4_ab_4m768.txt 2>&1; echo "probe rc $?"; grep -v amdgpu.ids $O/r5h_exact_v4_ab_4m768.txt
ROWS=2000000
FLOPS=$(python -c "print(4 * 2.0 * 1024 * $ROWS * 768)")
: > $O/r5h_exact_tile_kernel_only.txt
for cfg in "1 2 0" "1 4 0" "1 5 0" "1 5 8" "0 5 0" "0 5 1" "0 5 33" "0 5 2" "0 5 35" "0 5 39"; do
  set -- $cfg
  rm -rf /tmp/prof_x
  (cd /tmp && VSS_EXACT_FILTER=$1 VSS_EXACT_KERNEL=$2 VSS_EXACT_PROBE=$3 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_x -o x -- python $GRAFT_REPO_ROOT/tools/gpu_exact_probe.py $ROWS > /tmp/prof_x.log 2>&1)
  echo "filter $1 kernel $2 probe $3: $(python tools/rocprof_kernel_table.py /tmp/prof_x k_exact_scores $FLOPS)" >> $O/r5h_exact_tile_kernel_only.txt
done
cat $O/r5h_exact_tile_kernel_only.txt
