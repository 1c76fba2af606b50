#!/bin/bash
# round 5, session J: the rest of the -m gpu suite (session I stopped at a test that read a key the compact bench line no longer
# carried), then shader-clock ticks per phase of an expansion at limits of 257-512: 16 waves / plain order against 12 waves /
# pipelined (-DVSS_PHASE_TIMERS build).
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
(time timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "one_rank_rccl or properties_at or build_progress or two_rank") > $O/r5j_pytest_a.txt 2>&1; echo "pytest a rc $?"; tail -n 3 $O/r5j_pytest_a.txt
(time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider --ignore tests/test_gpu_parity.py --ignore tests/test_gpu_configs.py) > $O/r5j_pytest_b.txt 2>&1; echo "pytest b rc $?"; tail -n 3 $O/r5j_pytest_b.txt
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 600 python tools/gpu_wide_list_probe.py 10000000 768 cosine 16 128 10 512,288 > $O/r5j_wide_lists_phase_ticks_10m768.txt 2>&1; echo "probe 768 rc $?"; grep -v "^built\|amdgpu.ids" $O/r5j_wide_lists_phase_ticks_10m768.txt
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 600 python tools/gpu_wide_list_probe.py 3000000 1536 ip 32 128 100 480 > $O/r5j_wide_lists_phase_ticks_3m1536.txt 2>&1; echo "probe 1536 rc $?"; grep -v "^built\|amdgpu.ids" $O/r5j_wide_lists_phase_ticks_3m1536.txt
