#!/bin/bash
# round 4: two 8-wave workgroups per compute unit (2 walkers + 6 scoring waves each) against one 16-wave workgroup (4 + 12):
# does a workgroup of the next launch moving in beside a draining one pay?  development size (3M x 768), the driver's step counts
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
run() { echo "== $@"; env "$@" timeout 300 python bench.py --rows 3000000 --extras none --no-cpu-baseline --host-api-seconds 0 --no-small-launches --heldout-batches 1 --ef 64 --steps 20 --warmup 5 --regimes 1x1,16x1,8x3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('   20 steps: %.0f q/s, frac/launch %.3f, over wall %.3f' % (d['value'], r['frac'], r['frac_over_wall']))
for g in r['regimes']: print('   %dx%d: %.0f q/s, launch %.3f ms, frac/launch %.3f, over wall %.3f' % (g['batches_per_launch'], g['launches_in_flight'], g['queries_per_s'], g['avg_kernel_ms'], g['frac_per_launch'], g['frac_over_wall']))
"; }
run A=default 2>&1 | tee $O/r4_wgs_per_cu.txt
run VSS_SEARCH_WAVES=8 VSS_SEARCH_WALKERS=2 VSS_SEARCH_WGS_PER_CU=2 2>&1 | tee -a $O/r4_wgs_per_cu.txt
run VSS_SEARCH_WAVES=8 VSS_SEARCH_WALKERS=3 VSS_SEARCH_WGS_PER_CU=2 VSS_HASH_LDS_MAX_LOG2=13 2>&1 | tee -a $O/r4_wgs_per_cu.txt
