#!/bin/bash
# round 3, GPU session N: RowTouch (rows of the cached lists pulled into L2 one expansion ahead) in the one-query probe
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r3n_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3n_pytest.txt
timeout 300 python bench.py --config c2 --steps 4000 > $O/r3n_c2_touch.json 2> $O/r3n_c2_touch.err; echo "c2 touch rc $?"
VSS_SEARCH_TOUCH_ROWS=0 timeout 300 python bench.py --config c2 --steps 4000 --no-cpu-baseline > $O/r3n_c2_notouch.json 2> $O/r3n_c2_notouch.err; echo "c2 no-touch rc $?"
VSS_SEARCH_TOUCH_ROWS=0 VSS_PROBE_FLAG_WAIT=0 timeout 300 python bench.py --config c2 --steps 4000 --no-cpu-baseline > $O/r3n_c2_neither.json 2> $O/r3n_c2_neither.err; echo "c2 neither rc $?"
python - <<'PY'
import json
for f in ("r3n_c2_touch", "r3n_c2_notouch", "r3n_c2_neither"):
    try:
        r = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "q/s %.0f" % r["value"], "us/call %.1f" % (r["ms_per_step"] * 1e3), "kernel us %.1f" % (r["roofline"]["avg_kernel_ms"] * 1e3),
              "stream-wait us/call %.1f" % r["roofline"].get("us_per_call_waiting_on_the_stream", 0), "cpu", (r.get("cpu_baseline") or {}).get("value"),
              "agree", (r.get("cpu_baseline") or {}).get("agreement", {}).get("id_match_frac"))
    except Exception as e:
        print(f, "unreadable", e)
PY
