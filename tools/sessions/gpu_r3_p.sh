#!/bin/bash
# round 3, GPU session P: teams in the solo shape (3 helper waves score a share of every expansion's rows), with / without RowTouch
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r3p_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3p_pytest.txt
for team in 1 0; do for touch in 0 1; do
  VSS_SEARCH_TEAM=$team VSS_SEARCH_TOUCH_ROWS=$touch timeout 300 python bench.py --config c2 --steps 4000 --no-cpu-baseline > $O/r3p_c2_team${team}_touch${touch}.json 2> $O/r3p_c2_team${team}_touch${touch}.err; echo "c2 team $team touch $touch rc $?"
  echo "VSS_SEARCH_TEAM=$team VSS_SEARCH_TOUCH_ROWS=$touch" | tee -a $O/r3p_phase.txt
  VSS_SEARCH_TEAM=$team VSS_SEARCH_TOUCH_ROWS=$touch VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so timeout 300 python tools/gpu_solo_phase_probe.py 1000000 128 l2sq 16 128 64 2>&1 | grep -v amdgpu | grep "solo  *B=   1" | tee -a $O/r3p_phase.txt
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3p_c2_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "q/s %.0f" % r["value"], "us/call %.1f" % (r["ms_per_step"] * 1e3), "kernel us %.1f" % (r["roofline"]["avg_kernel_ms"] * 1e3),
              "stream-wait us/call %.1f" % r["roofline"].get("us_per_call_waiting_on_the_stream", 0))
    except Exception as e:
        print(f, "unreadable", e)
PY
