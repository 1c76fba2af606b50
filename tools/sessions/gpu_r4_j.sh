#!/bin/bash
# round 4, GPU session J: walkers per workgroup (4 / 6 / 8) and where the visited sets live, at thin expansions (large ef)
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/gpu_walkers_probe.py 3000000 768 cosine 32 256 10 60,128,256 2>&1 | grep -v amdgpu | tee $O/r4j_walkers_3m768.txt
timeout 500 python tools/gpu_walkers_probe.py 12500000 1536 ip 32 128 100 192,384,480 2>&1 | grep -v amdgpu | tee $O/r4j_walkers_12m1536.txt
