#!/bin/bash
# GPU session 6: engine shape sweep at the benchmark size, phase timers, and the tests fixed since session 5
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
rm -f $O/s6_*.txt
timeout 300 python tests/gpu_engine_probe.py 10000000 768 cosine 32 256 96 > $O/s6_engine_10m768.txt 2>&1
VSS_LIBRARY=$R/duckdb-vss_amd/libvssgpu_prof.so timeout 150 python tests/gpu_phase_probe.py 1000000 768 cosine 32 256 96 > $O/s6_phase_1m768.txt 2>&1
timeout 400 python -m pytest tests/test_gpu_parity2.py tests/test_host_harness.py -q -m gpu > $O/s6_tests.txt 2>&1
echo "tests rc=$?" >> $O/s6_tests.txt
cat $O/s6_engine_10m768.txt; tail -12 $O/s6_phase_1m768.txt; tail -8 $O/s6_tests.txt
