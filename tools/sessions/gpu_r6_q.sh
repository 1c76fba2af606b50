#!/bin/bash
# round 6, session Q: where the headline regime loses time with visited sets of twice the cells (session P) — phase ticks.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
for per in 0 128; do
  PROBE_PER_LIMIT=$per VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 600 python tools/gpu_wide_list_probe.py 3000000 768 cosine 32 256 10 60 > $O/r6q_ticks_per_limit_$per.txt 2>&1; echo "per $per rc $?"
  grep -v "^built\|amdgpu.ids" $O/r6q_ticks_per_limit_$per.txt | grep -A1 "retry in place" | cut -c1-330
done
