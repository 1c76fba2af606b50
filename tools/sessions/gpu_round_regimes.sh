#!/bin/bash
# gated launches — parity tests of the begin/end paths, then the regimes
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_parity.py tests/test_host_harness.py -q -m gpu -x -k "several_batches or pipelined or two_rank or register_queue or load_validates or harness or fuzz or concurrent" > $O/s13_tests.txt 2>&1; echo "pytest rc $?"
tail -n 6 $O/s13_tests.txt
timeout 600 python bench.py --no-cpu-baseline --host-api-seconds 0 --coalesce 8 --pipeline 2 --regimes 8x2u,8x1,4x2,4x1,8x3,2x2,1x3 > $O/s13_bench.json 2> $O/s13_bench.err; echo "bench rc $?"
tail -n 3 $O/s13_bench.err
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
d = json.loads(open(O + "/s13_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("timed: %dx%d gated value %.0f ms/step %.3f frac/launch %.3f kernel ms %.3f over wall %.3f recall %s" % (
    d["config"]["batches_per_launch"], d["config"]["launches_in_flight"], d["value"], d["ms_per_step"], r["frac"], r["avg_kernel_ms"],
    r["frac_over_wall"], d["recall_at_10"]))
for g in r["regimes"]:
    print("  %dx%d%s: %.0f q/s, %.3f ms/step, launch %.3f ms, frac/launch %.3f, over wall %.3f" % (
        g["batches_per_launch"], g["launches_in_flight"], "" if g["gated"] else "u", g["queries_per_s"], g["ms_per_step"],
        g["avg_kernel_ms"], g["frac_per_launch"], g["frac_over_wall"]))
PY
