#!/bin/bash
# GPU session: new collected tests + phase probes + team-pass microbenchmark (scratch output under gpurun_out/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_configs.py -q -m gpu -s --durations=15 > $O/s1_tests.txt 2>&1
echo "tests rc=$?" >> $O/s1_tests.txt
timeout 120 tests/microbench/team_pass_bench 1000000 > $O/s1_team_pass_3gb.txt 2>&1
export VSS_LIBRARY=$R/duckdb-vss_amd/libvssgpu_prof.so
timeout 200 python tests/gpu_phase_probe.py 1000000 128 l2sq 16 128 64 > $O/s1_phase_1m128.txt 2>&1
timeout 200 python tests/gpu_phase_probe.py 1000000 768 cosine 32 256 96 > $O/s1_phase_1m768.txt 2>&1
timeout 400 python tests/gpu_phase_probe.py 10000000 768 cosine 32 256 96 > $O/s1_phase_10m768.txt 2>&1
tail -5 $O/s1_tests.txt
