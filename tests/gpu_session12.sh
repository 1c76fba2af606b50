#!/bin/bash
# round-2 session 12: several batches per launch (parity + regimes), register queue for tombstone searches, whole suite
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
rm -f $O/config_tests.txt
timeout 900 python -m pytest tests -q -m gpu -x --durations=6 > $O/s12_tests.txt 2>&1; echo "pytest rc $?"
tail -n 12 $O/s12_tests.txt
cat $O/config_tests.txt
timeout 600 python bench.py --no-cpu-baseline --host-api-seconds 0 --regimes 4x1,8x1,8x2,4x3,2x3,2x2 > $O/s12_bench.json 2> $O/s12_bench.err; echo "bench rc $?"
tail -n 3 $O/s12_bench.err
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
d = json.loads(open(O + "/s12_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("timed: %dx%d value %.0f ms/step %.3f frac/launch %.3f kernel ms %.3f over wall %.3f recall %s" % (
    d["config"]["batches_per_launch"], d["config"]["launches_in_flight"], d["value"], d["ms_per_step"], r["frac"], r["avg_kernel_ms"],
    r["frac_over_wall"], d["recall_at_10"]))
for g in r["regimes"]:
    print("  %dx%d: %.0f q/s, %.3f ms/step, launch %.3f ms, frac/launch %.3f, over wall %.3f" % (
        g["batches_per_launch"], g["launches_in_flight"], g["queries_per_s"], g["ms_per_step"], g["avg_kernel_ms"],
        g["frac_per_launch"], g["frac_over_wall"]))
PY
