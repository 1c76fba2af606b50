#!/bin/bash
# GPU session 2: the search engine (persistent workgroups, walkers + scoring waves) — parity suite, fuzz diagnostics, shape sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -k "not full_benchmark_size and not config4" --durations=10 > $O/s2_tests.txt 2>&1
echo "tests rc=$?" >> $O/s2_tests.txt
( timeout 120 python tests/gpu_option_fuzz.py 10 1; timeout 120 python tests/gpu_option_fuzz.py 102 1 degenerate; timeout 120 python tests/gpu_option_fuzz.py 106 1 degenerate ) > $O/s2_fuzz_diag.txt 2>&1
timeout 400 python tests/gpu_engine_probe.py 10000000 768 cosine 32 256 96 > $O/s2_engine_10m768.txt 2>&1
export VSS_LIBRARY=$R/duckdb-vss_amd/libvssgpu_prof.so
timeout 200 python tests/gpu_phase_probe.py 1000000 768 cosine 32 256 96 > $O/s2_phase_1m768.txt 2>&1
tail -15 $O/s2_tests.txt
