#!/bin/bash
# round 3, GPU session U: shader-clock ticks per phase on queries the caches have not seen — teams / touches on and off
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
run() { echo "$@" | tee -a $O/r3u_phase.txt; env "$@" VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 300 python tools/gpu_solo_phase_probe.py 1000000 128 l2sq 16 128 64 2>&1 | grep -v amdgpu | grep "solo" | tee -a $O/r3u_phase.txt; }
run A=shipped
run VSS_SEARCH_TOUCH_ROWS=0
run VSS_SEARCH_TOUCH_ROWS=0 VSS_SEARCH_TOUCH_LISTS=0
run VSS_SEARCH_TEAM=0
run VSS_SEARCH_TEAM=0 VSS_SEARCH_TOUCH_ROWS=0 VSS_SEARCH_TOUCH_LISTS=0
