#!/bin/bash
# round 3, GPU session O: why RowTouch gains nothing — list-cache hit rate and phase ticks with / without the touches
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
for t in 1 0; do
  echo "VSS_SEARCH_TOUCH_ROWS=$t" | tee -a $O/r3o_phase.txt
  VSS_SEARCH_TOUCH_ROWS=$t VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 300 python tools/gpu_solo_phase_probe.py 1000000 128 l2sq 16 128 64 2>&1 | grep -v amdgpu | grep "B=   1" | tee -a $O/r3o_phase.txt
done
