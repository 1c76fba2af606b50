#!/bin/bash
# round 4, GPU session C: asynchronous list requests, LDS-only barriers, list touches by the crew — parity suite (without the
# three full-size tests), launch latency of the four engine variants at 3M x 768, phase ticks, and a development-size run of
# bench.py with two extras (orchestration check)
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r4c_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 6 $O/r4c_pytest.txt
timeout 400 python tools/gpu_crew_probe.py 3000000 768 cosine 32 256 80 2>&1 | grep -v amdgpu | tee $O/r4c_crew_probe_3m768.txt
VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 400 python tools/gpu_crew_probe.py 3000000 768 cosine 32 256 80 2>&1 | grep -v amdgpu | tee $O/r4c_crew_probe_3m768_phase_ticks.txt
(time timeout 600 python bench.py --rows 1000000 --dim 768 --steps 20 --warmup 5 --cpu-seconds 4 --extras c2,a13 --host-api-seconds 1) > $O/r4c_bench_dev.json 2> $O/r4c_bench_dev.err; echo "dev bench rc $?"; tail -c 1500 $O/r4c_bench_dev.err
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
try:
    d = json.loads([l for l in open(O + "/r4c_bench_dev.json") if l.startswith("{")][-1])
    print({k: d[k] for k in ("value", "recall_at_10", "recall_at_10_se", "ef_search", "small_launches", "extras")})
    print("agreement", d["cpu_baseline"]["agreement"])
    for name in d["extras"]["configs"]:
        e = d[name]
        print(name, {k: e.get(k) for k in ("error", "value", "unit", "wall_s", "exit_code")}, (e.get("roofline") or {}).get("frac"),
              ((e.get("cpu_baseline") or {}).get("agreement") or {}).get("unexplained_mismatches"))
except Exception as e:
    print("unreadable:", repr(e))
PY
