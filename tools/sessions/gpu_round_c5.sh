#!/bin/bash
# the configs[4] one-shard line
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 48 python bench.py --config c5 --rows ${C5_ROWS:-12500000} --steps 32 --warmup 16 > gpurun_out/s18_bench_c5.json 2> gpurun_out/s18_bench_c5.err; echo "c5 rc $?"
tail -c 1800 gpurun_out/s18_bench_c5.json; tail -n 3 gpurun_out/s18_bench_c5.err | cut -c1-300
