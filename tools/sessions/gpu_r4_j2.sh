#!/bin/bash
# round 4, GPU session J2: smaller visited sets that stay in LDS at large ef (overflowing queries are re-run), wall clock incl. retries
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
PROBE_TABLES=1 timeout 300 python tools/gpu_walkers_probe.py 3000000 768 cosine 32 256 10 128,256 2>&1 | grep -v amdgpu | tee $O/r4j2_tables_3m768.txt
PROBE_TABLES=1 timeout 500 python tools/gpu_walkers_probe.py 12500000 1536 ip 32 128 100 192,384,480 2>&1 | grep -v amdgpu | tee $O/r4j2_tables_12m1536.txt
