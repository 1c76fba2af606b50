#!/bin/bash
# round 3, GPU session Q: teams + ListTouch / RowTouch combinations, 4 and 8 waves per query
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r3q_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3q_pytest.txt
run() { # name lib-suffix env...
  local name=$1 suf=$2; shift 2
  env "$@" VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu$suf.so timeout 300 python bench.py --config c2 --steps 4000 --no-cpu-baseline > $O/r3q_c2_$name.json 2> $O/r3q_c2_$name.err; echo "c2 $name rc $?"
  local psuf=_prof; [ "$suf" = "_t8" ] && psuf=_prof8
  echo "$name" | tee -a $O/r3q_phase.txt
  env "$@" VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu$psuf.so timeout 300 python tools/gpu_solo_phase_probe.py 1000000 128 l2sq 16 128 64 2>&1 | grep -v amdgpu | grep "solo  *B=   1" | tee -a $O/r3q_phase.txt
}
run t4_lists1_rows0 "" VSS_SEARCH_TOUCH_LISTS=1 VSS_SEARCH_TOUCH_ROWS=0
run t4_lists1_rows1 "" VSS_SEARCH_TOUCH_LISTS=1 VSS_SEARCH_TOUCH_ROWS=1
run t4_lists0_rows0 "" VSS_SEARCH_TOUCH_LISTS=0 VSS_SEARCH_TOUCH_ROWS=0
run t8_lists1_rows0 _t8 VSS_SEARCH_TOUCH_LISTS=1 VSS_SEARCH_TOUCH_ROWS=0
run t8_lists1_rows1 _t8 VSS_SEARCH_TOUCH_LISTS=1 VSS_SEARCH_TOUCH_ROWS=1
run t8_lists0_rows0 _t8 VSS_SEARCH_TOUCH_LISTS=0 VSS_SEARCH_TOUCH_ROWS=0
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3q_c2_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "q/s %.0f" % r["value"], "us/call %.1f" % (r["ms_per_step"] * 1e3), "kernel us %.1f" % (r["roofline"]["avg_kernel_ms"] * 1e3),
              "stream-wait us/call %.1f" % r["roofline"].get("us_per_call_waiting_on_the_stream", 0))
    except Exception as e:
        print(f, "unreadable", e)
PY
