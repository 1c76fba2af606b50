#!/bin/bash
# round 6, session I: host-pointer searches of up to 256 queries (the HNSW_INDEX_JOIN chunk) take the pinned zero-copy path the
# one-query probe has had since round 3 — parity tests of the host-pointer entry points, the crew probe's per-call figures, the
# headline's small-launch lines.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --ignore tests/test_gpu_configs.py) > $O/r6i_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 6 $O/r6i_pytest.txt | cut -c1-400
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 400 python tools/gpu_crew_probe.py 3000000 768 cosine 32 256 80 2>&1 | grep -v amdgpu > $O/r6i_crew_probe_3m768_prof.txt; echo "crew probe rc $?"
grep "per call" $O/r6i_crew_probe_3m768_prof.txt | grep "crews+pipe plain" | cut -c1-200
(time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --extras none --cpu-seconds 4 --sidecar $O/r6i_c3_sidecar.json) > $O/r6i_c3.jsonl 2> $O/r6i_c3.err; grep '"detail": "regime"\|small_launches\|"detail": "repeat"\|host_api' $O/r6i_c3.jsonl | cut -c1-400; tail -n 1 $O/r6i_c3.jsonl | cut -c1-600
