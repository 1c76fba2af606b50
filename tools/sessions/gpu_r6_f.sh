#!/bin/bash
# round 6, session F: the pipelined level search's pick no longer looks for the best unexpanded entry on the critical path (it is
# found before the wait, together with the look-ahead's list requests) — parity suite, crew probe (one vss_search, the 204-query
# chunk), phase ticks at limits of 257-512, the configs[1] / configs[4]-shard lines and the headline without extras.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --ignore tests/test_gpu_configs.py) > $O/r6f_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 6 $O/r6f_pytest.txt | cut -c1-400
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 400 python tools/gpu_crew_probe.py 3000000 768 cosine 32 256 80 2>&1 | grep -v amdgpu > $O/r6f_crew_probe_3m768_prof.txt; echo "crew probe rc $?"
grep -A3 "^B=   1 \|^B= 204 " $O/r6f_crew_probe_3m768_prof.txt | grep "crews+pipe plain:\|^B=" | cut -c1-420
grep "per call" $O/r6f_crew_probe_3m768_prof.txt | grep "crews+pipe plain" | cut -c1-200
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 600 python tools/gpu_wide_list_probe.py 10000000 768 cosine 16 128 10 512,288 > $O/r6f_wide_lists_phase_ticks_10m768_prof.txt 2>&1; echo "wide probe rc $?"
grep -v "^built\|amdgpu.ids" $O/r6f_wide_lists_phase_ticks_10m768_prof.txt | grep -A1 "retry in place" | cut -c1-330
(time timeout 400 python bench.py --config c2 --steps 2000 --cpu-seconds 3 --sidecar $O/r6f_c2_sidecar.json) > $O/r6f_c2.jsonl 2> $O/r6f_c2.err; tail -n 1 $O/r6f_c2.jsonl | cut -c1-700
(time timeout 900 python bench.py --config c5 --steps 32 --warmup 16 --cpu-seconds 4 --sidecar $O/r6f_c5_sidecar.json) > $O/r6f_c5.jsonl 2> $O/r6f_c5.err; tail -n 1 $O/r6f_c5.jsonl | cut -c1-1200
(time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --extras none --cpu-seconds 4 --sidecar $O/r6f_c3_sidecar.json) > $O/r6f_c3.jsonl 2> $O/r6f_c3.err; grep '"detail": "regime"\|small_launches\|"detail": "repeat"' $O/r6f_c3.jsonl | cut -c1-400; tail -n 1 $O/r6f_c3.jsonl | cut -c1-1500
