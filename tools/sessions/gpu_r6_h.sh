#!/bin/bash
# round 6, session H: limits of 257-512 with MORE walkers per 12-wave workgroup over smaller compact visited sets (2^13 cells in
# 16 KiB instead of 2^14 in 32 KiB; a set that outgrows them moves to HBM) — the walkers' serial part (pick + gather + hand-over)
# is what a compute unit waits for at these limits since the accept phase went into the shadow of the row loads.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
PROBE_WALKERS=4:13,5:12,6:12,7:12 VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 900 python tools/gpu_wide_list_probe.py 10000000 768 cosine 16 128 10 512,384,288 > $O/r6h_wide_lists_walkers_10m768_prof.txt 2>&1; echo "wide probe rc $?"
grep -v "^built\|amdgpu.ids" $O/r6h_wide_lists_walkers_10m768_prof.txt | cut -c1-330
