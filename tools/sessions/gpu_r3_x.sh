#!/bin/bash
# round 3, GPU session X: rows in flight per team wave (2 shipped, 4) at M = 16 (32-row lists) and M = 32 (64-row lists), 128 dims
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
for suf in "" _r4; do
  echo "lib '$suf'" | tee -a $O/r3x_team_r.txt
  VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu$suf.so timeout 300 python bench.py --config c2 --steps 4000 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 M=16: %.0f queries/s %.1f us/call' % (r['value'], r['ms_per_step']*1e3))" | tee -a $O/r3x_team_r.txt
  VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu$suf.so timeout 300 python tools/gpu_team_probe.py 1000000 128 l2sq 32 128 64 2>&1 | grep -v amdgpu | tee -a $O/r3x_team_r.txt
done
