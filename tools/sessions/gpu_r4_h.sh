#!/bin/bash
# round 4, GPU session H: the exact path with the select folded into the score tile (parity tests incl. the overflow fall-back,
# timing against the plain path)
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "exact") > $O/r4h_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 5 $O/r4h_pytest.txt
timeout 400 python tools/gpu_exact_filter_probe.py 4000000 2>&1 | grep -v amdgpu | tee $O/r4h_exact_filter_probe.txt
