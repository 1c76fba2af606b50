#!/bin/bash
# GPU session 7: batched list merge + adaptive claims — parity first, then the shape sweep and phase timers
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
rm -f $O/s7_*.txt
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py tests/test_host_harness.py tests/test_gpu_configs.py -q -m gpu -s -k "not full_benchmark_size and not config4" > $O/s7_tests.txt 2>&1
echo "tests rc=$?" >> $O/s7_tests.txt
timeout 300 python tests/gpu_engine_probe.py 10000000 768 cosine 32 256 96 > $O/s7_engine_10m768.txt 2>&1
VSS_LIBRARY=$R/duckdb-vss_amd/libvssgpu_prof.so timeout 150 python tests/gpu_phase_probe.py 1000000 768 cosine 32 256 96 > $O/s7_phase_1m768.txt 2>&1
grep -E "passed|failed|configs\[1\]|FAILED|query " $O/s7_tests.txt | tail -12; cat $O/s7_engine_10m768.txt; tail -12 $O/s7_phase_1m768.txt
