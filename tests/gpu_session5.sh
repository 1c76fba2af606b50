#!/bin/bash
# GPU session 5: relay fix check (bounded steps), then the parity suites if the engine answers correctly
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
rm -f $O/s5_*.txt
timeout 60 tests/microbench/mailbox_test > $O/s5_mailbox.txt 2>&1
export VSS_LIBRARY=$R/duckdb-vss_amd/libvssgpu_dbg.so
for cfg in "2 1 1" "16 4 64"; do
  timeout 40 python tests/gpu_engine_trace.py $cfg >> $O/s5_trace.txt 2>&1
done
unset VSS_LIBRARY
timeout 120 python tests/gpu_engine_debug.py > $O/s5_debug_product.txt 2>&1
if grep -q "waves 16 walkers 0 nq 300: equal to oracle True" $O/s5_debug_product.txt; then
  ( timeout 100 python tests/gpu_option_fuzz.py 10 1; timeout 100 python tests/gpu_option_fuzz.py 102 1 degenerate; timeout 100 python tests/gpu_option_fuzz.py 106 1 degenerate ) > $O/s5_fuzz_diag.txt 2>&1
  timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py tests/test_host_harness.py tests/test_reference_sql_scenarios.py -q -m gpu -k "not full_benchmark_size" > $O/s5_tests.txt 2>&1
  echo "tests rc=$?" >> $O/s5_tests.txt
fi
grep -c "wrong results 0" $O/s5_mailbox.txt; cat $O/s5_trace.txt $O/s5_debug_product.txt; tail -15 $O/s5_tests.txt 2>/dev/null
