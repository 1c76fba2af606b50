#!/bin/bash
# the multi-batch parity test and the driver's bench command after raising the batches-per-launch limit to 16
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 120 python -m pytest tests/test_gpu_parity2.py -q -m gpu -x -p no:cacheprovider -k "several_batches" 2>&1 | tail -n 1
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --ef 96 --regimes 8x3 --no-cpu-baseline --host-api-seconds 0 > $O/s17_bench.json 2> $O/s17_bench.err; echo "bench rc $?"
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
d = json.loads([l for l in open(O + "/s17_bench.json").read().splitlines() if l.startswith("{")][-1])
r = d["roofline"]
print("%dx%d steps %d: value %.0f ms/step %.3f frac/launch %.3f kernel ms %.3f (%d launches) over wall %.3f recall %s traffic %s" % (
    d["config"]["batches_per_launch"], d["config"]["launches_in_flight"], d["steps"], d["value"], d["ms_per_step"], r["frac"], r["avg_kernel_ms"], r["launches"], r["frac_over_wall"], d["recall_at_10"], r["traffic_source"]))
for g in r["regimes"]:
    print("  %dx%d%s: %.0f q/s, launch %.3f ms, frac/launch %.3f, over wall %.3f" % (g["batches_per_launch"], g["launches_in_flight"], "" if g["gated"] else "u", g["queries_per_s"], g["avg_kernel_ms"], g["frac_per_launch"], g["frac_over_wall"]))
PY
