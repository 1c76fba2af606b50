#!/bin/bash
# round 6, session C: the accept loop without the radius in its loop-carried chain (WaveList::place_finite), two neighbour lists
# in flight next to the 8-register list, and the visited set's probe with a plain read ahead of the compare-and-swap
# (-DVSS_VISITED_READ_FIRST: libvssgpu_rf*.so) — microbenchmark, parity subset on both libraries, phase ticks in the three
# regimes; then the co-resident 8-shard build of configs[3] with 4 (HIP's default) and 8 hardware queues.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
(timeout 300 tools/microbench/walker_ops) > $O/r6c_walker_ops.txt 2>&1; echo "walker_ops rc $?"; grep "round 6\|identical\|visited" $O/r6c_walker_ops.txt | cut -c1-230
(timeout 300 tools/microbench/walker_ops_read_first) > $O/r6c_walker_ops_read_first.txt 2>&1; echo "walker_ops_read_first rc $?"; grep "visited" $O/r6c_walker_ops_read_first.txt | cut -c1-230
(time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -m gpu -x -q -p no:cacheprovider \
   -k "reference_built or variants_agree or compact_visited or both_engine_shapes or fuzz or bulk_build or goldens or limits_beyond or tombstones or removed_slots") > $O/r6c_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r6c_pytest.txt | cut -c1-300
(time VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_rf.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -m gpu -x -q -p no:cacheprovider \
   -k "reference_built or variants_agree or compact_visited or both_engine_shapes or fuzz or limits_beyond or tombstones or removed_slots") > $O/r6c_pytest_rf.txt 2>&1; echo "pytest rf rc $?"; tail -n 3 $O/r6c_pytest_rf.txt | cut -c1-300
for lib in prof rf_prof; do
  VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_$lib.so timeout 400 python tools/gpu_crew_probe.py 3000000 768 cosine 32 256 80 2>&1 | grep -v amdgpu > $O/r6c_crew_probe_3m768_$lib.txt; echo "crew probe $lib rc $?"
  grep -A3 "^B=   1 \|^B= 204 " $O/r6c_crew_probe_3m768_$lib.txt | grep "crews+pipe plain:" | cut -c1-420
  grep "per launch\|per call" $O/r6c_crew_probe_3m768_$lib.txt | grep "crews+pipe plain" | cut -c1-200
done
for lib in prof rf_prof; do
  VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_$lib.so timeout 600 python tools/gpu_wide_list_probe.py 10000000 768 cosine 16 128 10 512,288 > $O/r6c_wide_lists_phase_ticks_10m768_$lib.txt 2>&1; echo "wide probe $lib rc $?"
  grep -v "^built\|amdgpu.ids" $O/r6c_wide_lists_phase_ticks_10m768_$lib.txt | grep -A1 "retry in place" | cut -c1-330
done
for lib in prof rf_prof; do
  VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_$lib.so timeout 300 python tools/gpu_solo_phase_probe.py 1000000 128 l2sq 16 128 64 > $O/r6c_solo_phase_1m128_$lib.txt 2>&1; echo "solo probe $lib rc $?"
  grep "^solo" $O/r6c_solo_phase_1m128_$lib.txt | cut -c1-330 | head -4
done
(time timeout 600 python bench.py --config c4 --build-only --extras none --sidecar $O/r6c_c4_build_4q.json) > $O/r6c_c4_build_4_queues.jsonl 2>&1; tail -n 1 $O/r6c_c4_build_4_queues.jsonl | cut -c1-400
(time GPU_MAX_HW_QUEUES=8 timeout 600 python bench.py --config c4 --build-only --extras none --sidecar $O/r6c_c4_build_8q.json) > $O/r6c_c4_build_8_queues.jsonl 2>&1; tail -n 1 $O/r6c_c4_build_8_queues.jsonl | cut -c1-400
