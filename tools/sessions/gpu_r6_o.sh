#!/bin/bash
# round 6, session O: the count of running walkers is read BEFORE the gather and used after it (crew switch, rows per claim): one LDS round
# trip in the gather's shadow instead of two on the serial path — parity suite and the probes of session L.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
if [ "$1" == "engine" ]; then
(time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --ignore tests/test_gpu_configs.py) > $O/r6o_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 6 $O/r6o_pytest.txt | cut -c1-400
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 400 python tools/gpu_crew_probe.py 3000000 768 cosine 32 256 80 2>&1 | grep -v amdgpu > $O/r6o_crew_probe_3m768_prof.txt; echo "crew probe rc $?"
grep -A3 "^B=   1 \|^B= 204 " $O/r6o_crew_probe_3m768_prof.txt | grep "crews+pipe plain:\|^B=" | cut -c1-420
grep "per call" $O/r6o_crew_probe_3m768_prof.txt | grep "crews+pipe plain" | cut -c1-200
grep "x 1024 queries per launch, crews+pipe plain" $O/r6o_crew_probe_3m768_prof.txt
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 300 python tools/gpu_solo_phase_probe.py 1000000 128 l2sq 16 128 64 > $O/r6o_solo_phase_1m128_prof.txt 2>&1; echo "solo probe rc $?"
grep -v "amdgpu.ids" $O/r6o_solo_phase_1m128_prof.txt | cut -c1-330 | head -8
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 600 python tools/gpu_wide_list_probe.py 10000000 768 cosine 16 128 10 512,288 > $O/r6o_wide_lists_phase_ticks_10m768_prof.txt 2>&1; echo "wide probe rc $?"
grep -v "^built\|amdgpu.ids" $O/r6o_wide_lists_phase_ticks_10m768_prof.txt | grep -A1 "retry in place" | cut -c1-330
(time timeout 400 python bench.py --config c2 --steps 2000 --cpu-seconds 3 --sidecar $O/r6o_c2_sidecar.json) > $O/r6o_c2.jsonl 2> $O/r6o_c2.err; tail -n 1 $O/r6o_c2.jsonl | cut -c1-300
(time timeout 900 python bench.py --config c5 --steps 32 --warmup 16 --cpu-seconds 4 --sidecar $O/r6o_c5_sidecar.json) > $O/r6o_c5.jsonl 2> $O/r6o_c5.err; tail -n 1 $O/r6o_c5.jsonl | cut -c1-1200
(time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --extras none --cpu-seconds 4 --sidecar $O/r6o_c3_sidecar.json) > $O/r6o_c3.jsonl 2> $O/r6o_c3.err; grep '"detail": "regime"\|small_launches\|"detail": "repeat"' $O/r6o_c3.jsonl | cut -c1-400; tail -n 1 $O/r6o_c3.jsonl | cut -c1-700
fi
