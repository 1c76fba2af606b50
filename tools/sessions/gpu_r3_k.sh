#!/bin/bash
# round 3, GPU session K: unrolled instantiations for 512 / 1024 dimensions — parity, then against the looping kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r3k_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3k_pytest.txt
for d in 512 1024; do
  timeout 300 python tools/gpu_dim_probe.py 2000000 $d 2>&1 | grep -v amdgpu | tee -a $O/r3k_dims.txt
  VSS_FORCE_LOOPING=1 timeout 300 python tools/gpu_dim_probe.py 2000000 $d 2>&1 | grep -v amdgpu | tee -a $O/r3k_dims.txt
done
