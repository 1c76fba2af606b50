#!/bin/bash
# round 6, session M: the list cache's requests are global loads again (their opaque pointer had made them FLAT: counted on
# lgkmcnt, so every wait for an LDS read behind one waited for its HBM round trip) — parity suite, crew / solo probes, c2.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --ignore tests/test_gpu_configs.py) > $O/r6m_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 6 $O/r6m_pytest.txt | cut -c1-400
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 400 python tools/gpu_crew_probe.py 3000000 768 cosine 32 256 80 2>&1 | grep -v amdgpu > $O/r6m_crew_probe_3m768_prof.txt; echo "crew probe rc $?"
grep -A3 "^B=   1 \|^B= 204 " $O/r6m_crew_probe_3m768_prof.txt | grep "crews+pipe plain:\|^B=" | cut -c1-420
grep "per call" $O/r6m_crew_probe_3m768_prof.txt | grep "crews+pipe plain" | cut -c1-200
grep "x 1024 queries per launch, crews+pipe plain" $O/r6m_crew_probe_3m768_prof.txt
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 300 python tools/gpu_solo_phase_probe.py 1000000 128 l2sq 16 128 64 > $O/r6m_solo_phase_1m128_prof.txt 2>&1; echo "solo probe rc $?"
grep -v "amdgpu.ids" $O/r6m_solo_phase_1m128_prof.txt | cut -c1-330 | head -8
(time timeout 400 python bench.py --config c2 --steps 2000 --cpu-seconds 3 --sidecar $O/r6m_c2_sidecar.json) > $O/r6m_c2.jsonl 2> $O/r6m_c2.err; tail -n 1 $O/r6m_c2.jsonl | cut -c1-300
