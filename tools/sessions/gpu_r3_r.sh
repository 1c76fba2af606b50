#!/bin/bash
# round 3, GPU session R: teams of 8 (default) with the helpers doing RowTouch; against 4 waves and against the touches off
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r3r_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3r_pytest.txt
run() { # name lib-suffix env...
  local name=$1 suf=$2; shift 2
  env "$@" VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu$suf.so timeout 300 python bench.py --config c2 --steps 4000 --no-cpu-baseline > $O/r3r_c2_$name.json 2> $O/r3r_c2_$name.err; echo "c2 $name rc $?"
}
run t8_default "" A=1
run t8_rows0 "" VSS_SEARCH_TOUCH_ROWS=0
run t8_rows0_lists0 "" VSS_SEARCH_TOUCH_ROWS=0 VSS_SEARCH_TOUCH_LISTS=0
run t4_default _t4 A=1
run t8_default_again "" A=1
run one_wave "" VSS_SEARCH_TEAM=0
timeout 300 python bench.py --config c2 --steps 4000 > $O/r3r_c2_full.json 2> $O/r3r_c2_full.err; echo "c2 full rc $?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3r_c2_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "q/s %.0f" % r["value"], "us/call %.1f" % (r["ms_per_step"] * 1e3), "kernel us %.1f" % (r["roofline"]["avg_kernel_ms"] * 1e3),
              "stream-wait us/call %.1f" % r["roofline"].get("us_per_call_waiting_on_the_stream", 0), "cpu", (r.get("cpu_baseline") or {}).get("value"),
              (r.get("cpu_baseline") or {}).get("agreement", {}).get("id_match_frac"))
    except Exception as e:
        print(f, "unreadable", e)
PY
