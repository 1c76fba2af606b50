#!/bin/bash
# round 5, session K: (1) the sorted insert that skips untouched list registers — exactness (collected tests of the list paths:
# build graph bytes, search variants, fuzz) and the wide-list probe with phase ticks again; (2) bulk build at the headline options
# (10M x 768, M 32, ef_construction 384 -> the 8-register list's phase-A kernel) with 3 waves per SIMD (shipped) against 4 waves per
# SIMD (128 registers, 40 bytes of scratch): libvssgpu_b4.so.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
(time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -m gpu -x -q -p no:cacheprovider \
   -k "bulk_build_graph or sequential_build or variants_agree or fuzz or compact_visited or limits_beyond or removed_slots or register_queue or reference_goldens") > $O/r5k_pytest.txt 2>&1
echo "pytest rc $?"; tail -n 4 $O/r5k_pytest.txt
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 600 python tools/gpu_wide_list_probe.py 10000000 768 cosine 16 128 10 512 > $O/r5k_wide_lists_phase_ticks_10m768.txt 2>&1; echo "probe 768 rc $?"; grep -v "^built\|amdgpu.ids" $O/r5k_wide_lists_phase_ticks_10m768.txt
for lib in libvssgpu.so libvssgpu_b4.so; do
  VSS_LIBRARY=$PWD/duckdb-vss_amd/$lib timeout 400 python bench.py --config c3 --build-only --extras none --sidecar $O/r5k_build_$lib.full.json > $O/r5k_build_$lib.jsonl 2> $O/r5k_build_$lib.err; echo "build $lib rc $?"
  tail -n 1 $O/r5k_build_$lib.jsonl | cut -c1-700
done
