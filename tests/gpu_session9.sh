#!/bin/bash
# GPU session 9: one expansion of look-ahead — parity, shape / look-ahead sweep, configs[1] line
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
rm -f $O/s9_*.txt $O/s9_*.json
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py tests/test_reference_sql_scenarios.py -q -m gpu -x -k "not full_benchmark_size" > $O/s9_tests.txt 2>&1
echo "tests rc=$?" >> $O/s9_tests.txt
timeout 300 python tests/gpu_engine_probe.py 10000000 768 cosine 32 256 96 > $O/s9_engine_10m768.txt 2>&1
timeout 120 python bench.py --config c2 --no-cpu-baseline > $O/s9_bench_c2.json 2> $O/s9_bench_c2.err
tail -6 $O/s9_tests.txt; cat $O/s9_engine_10m768.txt; cat $O/s9_bench_c2.json | cut -c1-400
