#!/bin/bash
# round 6, session E: the crew pulls the rows of the lists its walker holds ahead of their use into L2 (CREW_TOUCH_ROWS: launches
# of at most 32 queries) — parity subset, the > 2^24-slot test, the crew probe with the touches on and off (VSS_SEARCH_TOUCH_ROWS=0).
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
(time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -m gpu -x -q -p no:cacheprovider \
   -k "reference_built or variants_agree or compact_visited or both_engine_shapes or fuzz or config1 or concurrent or pipelined_contexts or host_pointer") > $O/r6e_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r6e_pytest.txt | cut -c1-300
(time timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -p no:cacheprovider -s -k "25_bit") > $O/r6e_pytest_25bit.txt 2>&1; echo "25-bit rc $?"; tail -n 6 $O/r6e_pytest_25bit.txt | cut -c1-500
for touch in 1 0; do
  VSS_SEARCH_TOUCH_ROWS=$touch VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 400 python tools/gpu_crew_probe.py 3000000 768 cosine 32 256 80 2>&1 | grep -v amdgpu > $O/r6e_crew_probe_3m768_touch_rows_$touch.txt; echo "crew probe touch_rows=$touch rc $?"
  grep -A3 "^B=   1 \|^B=   8 \|^B= 204 " $O/r6e_crew_probe_3m768_touch_rows_$touch.txt | grep "crews+pipe plain:\|^B=" | cut -c1-420
  grep "per call" $O/r6e_crew_probe_3m768_touch_rows_$touch.txt | grep "crews+pipe plain" | cut -c1-200
done
