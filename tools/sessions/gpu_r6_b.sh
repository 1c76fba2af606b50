#!/bin/bash
# round 6, session B: the blocked, right-aligned candidate list (csrc/wave_primitives.h WaveList) against round 5's lane-major
# one — the walker's operations in isolation (tools/microbench/walker_ops: cycles per accept phase / pick / look-ahead, final lists
# compared entry by entry), the parity suite on the new list, shader-clock ticks per phase before (libvssgpu_r5_prof.so = the
# tree before the change) and after in the three regimes the review names: one query / a 204-query chunk at 768 dims (crew),
# limits of 257-512 at 768 dims, one query at 1M x 128 (team shape) — and the quality study at 1M rows for the default options.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
(time timeout 300 tools/microbench/walker_ops) > $O/r6b_walker_ops.txt 2>&1; echo "walker_ops rc $?"; cat $O/r6b_walker_ops.txt | cut -c1-250
(time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --ignore tests/test_gpu_configs.py) > $O/r6b_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 6 $O/r6b_pytest.txt | cut -c1-400
for lib in r5_prof prof; do
  VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_$lib.so timeout 400 python tools/gpu_crew_probe.py 3000000 768 cosine 32 256 80 2>&1 | grep -v amdgpu > $O/r6b_crew_probe_3m768_$lib.txt; echo "crew probe $lib rc $?"
  grep -A3 "^B=   1 \|^B= 204 " $O/r6b_crew_probe_3m768_$lib.txt | grep "crews+pipe plain:\|round 3 " | cut -c1-420
  grep "per launch\|per call" $O/r6b_crew_probe_3m768_$lib.txt | grep "crews+pipe plain" | cut -c1-200
done
for lib in r5_prof prof; do
  VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_$lib.so timeout 600 python tools/gpu_wide_list_probe.py 10000000 768 cosine 16 128 10 512,288 > $O/r6b_wide_lists_phase_ticks_10m768_$lib.txt 2>&1; echo "wide probe $lib rc $?"
  grep -v "^built\|amdgpu.ids" $O/r6b_wide_lists_phase_ticks_10m768_$lib.txt | cut -c1-330
done
for lib in r5_prof prof; do
  VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_$lib.so timeout 300 python tools/gpu_solo_phase_probe.py 1000000 128 l2sq 16 128 64 > $O/r6b_solo_phase_1m128_$lib.txt 2>&1; echo "solo probe $lib rc $?"
  grep -v "amdgpu.ids" $O/r6b_solo_phase_1m128_$lib.txt | cut -c1-330 | head -12
done
(time timeout 900 python bench.py --config quality --quality-rows 1000000 --quality-options 16/128 --quality-efs 64,128,256,512,1024,1536 \
   --sidecar $O/r6b_quality_1m_sidecar.json) > $O/r6b_quality_1m.jsonl 2> $O/r6b_quality_1m.err; tail -n 1 $O/r6b_quality_1m.jsonl | cut -c1-1500
