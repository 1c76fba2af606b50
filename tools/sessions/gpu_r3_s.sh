#!/bin/bash
# round 3, GPU session S: teams for every row width — parity, then launch latency per shape and batch size at 128 and 768 dims
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r3s_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3s_pytest.txt
timeout 300 python tools/gpu_team_probe.py 1000000 128 l2sq 16 128 64 2>&1 | grep -v amdgpu | tee $O/r3s_team_probe_1m128.txt
timeout 600 python tools/gpu_team_probe.py 3000000 768 cosine 32 256 80 2>&1 | grep -v amdgpu | tee $O/r3s_team_probe_3m768.txt
