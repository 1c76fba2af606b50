#!/bin/bash
# round 3, GPU session C: solo shape after the load-clustering fix — parity, timing, phase breakdown.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r3c_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3c_pytest.txt
timeout 300 python tools/gpu_solo_probe.py 1000000 128 l2sq 16 128 64 > $O/r3c_solo_1m128.txt 2>&1; echo "solo probe rc $?"; cat $O/r3c_solo_1m128.txt
VSS_LIBRARY=duckdb-vss_amd/libvssgpu_prof.so timeout 300 python tools/gpu_solo_phase_probe.py 1000000 128 l2sq 16 128 64 > $O/r3c_solo_phase_1m128.txt 2>&1; echo "phase probe rc $?"; cat $O/r3c_solo_phase_1m128.txt
