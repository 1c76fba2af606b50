#!/bin/bash
# round 4, GPU session A: crews in the workgroup engine — parity suite (without the three full-size tests), launch latency
# with and without crews at 3M x 768 (one query ... ten batches), phase ticks of an expansion (profiling build)
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r4a_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 6 $O/r4a_pytest.txt
timeout 400 python tools/gpu_crew_probe.py 3000000 768 cosine 32 256 80 2>&1 | grep -v amdgpu | tee $O/r4a_crew_probe_3m768.txt
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 400 python tools/gpu_crew_probe.py 3000000 768 cosine 32 256 80 2>&1 | grep -v amdgpu | tee $O/r4a_crew_probe_3m768_phase_ticks.txt
