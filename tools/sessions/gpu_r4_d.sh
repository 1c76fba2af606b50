#!/bin/bash
# round 4, GPU session D: crew refinements A/B (walker's SIMD spared, no list requests by the walker), the RCCL exchange behind
# the C ABI (one rank), parity of the engine variants
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "variants or sharded or host_harness or rccl or both_engine_shapes or several_batches") > $O/r4d_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 6 $O/r4d_pytest.txt
timeout 500 python tools/gpu_crew_probe.py 3000000 768 cosine 32 256 80 2>&1 | grep -v amdgpu | tee $O/r4d_crew_probe_3m768.txt
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 500 python tools/gpu_crew_probe.py 3000000 768 cosine 32 256 80 2>&1 | grep -v amdgpu | grep -A12 "^B=   1" | tee $O/r4d_crew_probe_3m768_phase_ticks.txt
