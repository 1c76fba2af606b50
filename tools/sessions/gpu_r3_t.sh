#!/bin/bash
# round 3, GPU session T: touches for launches of more than 8 queries (teams, 128 dims); the automatic choice
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
for tm in 8 32 256; do
  echo "VSS_SEARCH_TOUCH_MAX=$tm" | tee -a $O/r3t_team_probe_1m128.txt
  VSS_SEARCH_TOUCH_MAX=$tm timeout 300 python tools/gpu_team_probe.py 1000000 128 l2sq 16 128 64 2>&1 | grep -v amdgpu | tee -a $O/r3t_team_probe_1m128.txt
done
