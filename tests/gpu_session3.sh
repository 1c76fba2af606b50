#!/bin/bash
# GPU session 3: engine bring-up diagnostics (core dumps off: a faulting process must die fast)
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
VSS_LIBRARY=$R/duckdb-vss_amd/libvssgpu_dbg.so timeout 150 python tests/gpu_engine_debug.py > $O/s3_debug_paranoid.txt 2>&1
timeout 150 python tests/gpu_engine_debug.py > $O/s3_debug_product.txt 2>&1
( timeout 100 python tests/gpu_option_fuzz.py 10 1; timeout 100 python tests/gpu_option_fuzz.py 102 1 degenerate; timeout 100 python tests/gpu_option_fuzz.py 106 1 degenerate ) > $O/s3_fuzz_diag.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py tests/test_host_harness.py -q -m gpu -x -k "not full_benchmark_size" > $O/s3_tests.txt 2>&1
echo "tests rc=$?" >> $O/s3_tests.txt
tail -5 $O/s3_debug_paranoid.txt $O/s3_debug_product.txt $O/s3_tests.txt
