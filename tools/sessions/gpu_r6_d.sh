#!/bin/bash
# round 6, session D: the team shape running the pipelined level search (k_search_solo<.., 8> + TeamPool), the compact visited set
# reading first and MOVING to HBM when it outgrows its cells (VisitedSet::migrate), its 25-bit key form — whole parity suite,
# the > 2^24-slot test, phase ticks, then the configs[1] / configs[4]-shard / headline lines of bench.py.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --ignore tests/test_gpu_configs.py) > $O/r6d_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 6 $O/r6d_pytest.txt | cut -c1-400
(time timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -p no:cacheprovider -s -k "25_bit") > $O/r6d_pytest_25bit.txt 2>&1; echo "25-bit rc $?"; tail -n 8 $O/r6d_pytest_25bit.txt | cut -c1-500
(timeout 300 tools/microbench/walker_ops) > $O/r6d_walker_ops.txt 2>&1; echo "walker_ops rc $?"; grep "visited\|identical$" $O/r6d_walker_ops.txt | cut -c1-200
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 300 python tools/gpu_solo_phase_probe.py 1000000 128 l2sq 16 128 64 > $O/r6d_solo_phase_1m128_prof.txt 2>&1; echo "solo probe rc $?"
grep -v "amdgpu.ids" $O/r6d_solo_phase_1m128_prof.txt | cut -c1-330 | head -8
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 600 python tools/gpu_wide_list_probe.py 10000000 768 cosine 16 128 10 512,288 > $O/r6d_wide_lists_phase_ticks_10m768_prof.txt 2>&1; echo "wide probe rc $?"
grep -v "^built\|amdgpu.ids" $O/r6d_wide_lists_phase_ticks_10m768_prof.txt | grep -A1 "retry in place" | cut -c1-330
(time timeout 400 python bench.py --config c2 --steps 2000 --cpu-seconds 3 --sidecar $O/r6d_c2_sidecar.json) > $O/r6d_c2.jsonl 2> $O/r6d_c2.err; tail -n 1 $O/r6d_c2.jsonl | cut -c1-700
(time timeout 900 python bench.py --config c5 --steps 32 --warmup 16 --cpu-seconds 4 --sidecar $O/r6d_c5_sidecar.json) > $O/r6d_c5.jsonl 2> $O/r6d_c5.err; tail -n 1 $O/r6d_c5.jsonl | cut -c1-1200
(time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --extras none --cpu-seconds 4 --sidecar $O/r6d_c3_sidecar.json) > $O/r6d_c3.jsonl 2> $O/r6d_c3.err; grep '"detail": "regime"\|small_launches\|"detail": "repeat"' $O/r6d_c3.jsonl | cut -c1-400; tail -n 1 $O/r6d_c3.jsonl | cut -c1-1500
