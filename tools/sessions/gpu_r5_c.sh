#!/bin/bash
# round 5, session C: (1) limits of 257-512 — 12 waves + pipelined level search against round 4's 16 waves / plain order, rates at
# 768 and 1536 dims (session B's probe tripped over a stats call; its 32 collected tests were green); (2) what the f32 matrix
# pipe sustains (tools/microbench/mfma_f32_peak.hip).
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
tools/microbench/mfma_f32_peak > $O/r5c_mfma_f32_peak.txt 2>&1; echo "mfma rc $?"; cat $O/r5c_mfma_f32_peak.txt
timeout 600 python tools/gpu_wide_list_probe.py 10000000 768 cosine 16 128 10 512,384,288 > $O/r5c_wide_lists_10m768.txt 2>&1; echo "probe 768 rc $?"; grep -v "^built" $O/r5c_wide_lists_10m768.txt | tail -n 14
PROBE_EXTRA=1 timeout 600 python tools/gpu_wide_list_probe.py 3000000 1536 ip 32 128 100 480,320 > $O/r5c_wide_lists_3m1536.txt 2>&1; echo "probe 1536 rc $?"; grep -v "^built" $O/r5c_wide_lists_3m1536.txt | tail -n 18
