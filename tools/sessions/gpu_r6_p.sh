#!/bin/bash
# round 6, session P: the headline's visited sets at 128 cells per entry of the limit instead of 64 (limit 60: 8192 cells = 32 KiB per
# walker instead of 4096 — a query's 2150 visits fill the smaller table to 52 %), same box, same index options.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
for per in 0 128; do
  (VSS_VISITED_PER_LIMIT=$per timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --extras none --cpu-seconds 2 --no-cpu-baseline --host-api-seconds 0 --sidecar $O/r6p_c3_per_$per.json) > $O/r6p_c3_per_$per.jsonl 2> $O/r6p_c3_per_$per.err; echo "cells per limit $per rc $?"
  grep '"detail": "regime"\|small_launches\|"detail": "repeat"' $O/r6p_c3_per_$per.jsonl | cut -c1-330
  tail -n 1 $O/r6p_c3_per_$per.jsonl | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['avg_kernel_ms'], d['roofline'].get('visited_set'))"
done
