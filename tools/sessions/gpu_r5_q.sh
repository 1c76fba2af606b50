#!/bin/bash
# round 5, session Q: the filtered exact pass with its selects on a second stream — collected exact tests, then the A/B probe.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "exact or smoke") > $O/r5q_pytest_exact.txt 2>&1; echo "pytest rc $?"; tail -n 4 $O/r5q_pytest_exact.txt
timeout 600 python tools/gpu_exact_v3_probe.py 10000000 > $O/r5q_exact_overlap_10m768.txt 2>&1; echo "probe rc $?"; grep -v amdgpu.ids $O/r5q_exact_overlap_10m768.txt
