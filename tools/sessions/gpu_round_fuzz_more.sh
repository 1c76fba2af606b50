#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 400 python tests/gpu_option_fuzz.py 20 90 > gpurun_out/s16_fuzz_20_109.txt 2>&1; echo "fuzz rc $?"; tail -n 4 gpurun_out/s16_fuzz_20_109.txt | cut -c1-600
timeout 300 python tests/gpu_option_fuzz.py 108 40 degenerate > gpurun_out/s16_fuzz_degenerate_108_147.txt 2>&1; echo "degenerate fuzz rc $?"; tail -n 4 gpurun_out/s16_fuzz_degenerate_108_147.txt | cut -c1-600
