#!/bin/bash
# GPU session 8: the whole -m gpu suite (full sizes included), the exact path's rate, and the default bench line
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
rm -f $O/s8_*.txt
timeout 900 python -m pytest tests -q -m gpu --durations=8 > $O/s8_tests.txt 2>&1
echo "tests rc=$?" >> $O/s8_tests.txt
timeout 120 python tests/gpu_exact_probe.py 1000000 > $O/s8_exact.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline > $O/s8_bench.json 2> $O/s8_bench.err
timeout 120 python bench.py --config c2 > $O/s8_bench_c2.json 2> $O/s8_bench_c2.err
tail -14 $O/s8_tests.txt; cat $O/s8_exact.txt; cat $O/s8_bench.json | cut -c1-3000; tail -3 $O/s8_bench.err; cat $O/s8_bench_c2.json; tail -3 $O/s8_bench_c2.err
