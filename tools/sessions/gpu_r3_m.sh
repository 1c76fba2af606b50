#!/bin/bash
# round 3, GPU session M: the one-query probe waiting on a completion word (against the stream wait), and phase-B build variants
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r3m_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3m_pytest.txt
timeout 300 python bench.py --config c2 --steps 4000 > $O/r3m_c2_flag.json 2> $O/r3m_c2_flag.err; echo "c2 flag rc $?"
VSS_PROBE_FLAG_WAIT=0 timeout 300 python bench.py --config c2 --steps 4000 --no-cpu-baseline > $O/r3m_c2_stream.json 2> $O/r3m_c2_stream.err; echo "c2 stream rc $?"
python - <<'PY'
import json
for f in ("r3m_c2_flag", "r3m_c2_stream"):
    try:
        r = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "q/s %.0f" % r["value"], "us/call %.1f" % (r["ms_per_step"] * 1e3), "kernel us %.1f" % (r["roofline"]["avg_kernel_ms"] * 1e3),
              "stream-wait us/call %.1f" % r["roofline"].get("us_per_call_waiting_on_the_stream", 0), "cpu", (r.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "unreadable", e)
PY
for v in "" _pb4 _pb8w4 _pb4w4; do
  echo "phase B variant '$v'" | tee -a $O/r3m_phase_b.txt
  VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu$v.so timeout 300 python tools/gpu_build_probe.py 3000000 2>&1 | grep -v amdgpu | tee -a $O/r3m_phase_b.txt
done
