#!/bin/bash
# round 4: the default step count (128 steps = 8 launches of 16 batches) and the launch regimes of the final engine
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python bench.py --extras none --no-cpu-baseline --host-api-seconds 0 --regimes 4x1,8x1,16x1,8x2,8x3) > $O/r4_bench_default_128_steps.json 2> $O/r4_bench_default_128_steps.err; echo "rc $?"
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
d = json.loads([l for l in open(O + "/r4_bench_default_128_steps.json") if l.startswith("{")][-1])
r = d["roofline"]
print("128 steps: value %.0f q/s ms/step %.3f ef %d recall %.4f frac/launch %.3f over wall %.3f kernel %.3f ms x %d; small %s" % (d["value"], d["ms_per_step"], d["ef_search"], d["recall_at_10"],
      r["frac"], r["frac_over_wall"], r["avg_kernel_ms"], r["launches"], {k: round(v["us_per_call"], 1) for k, v in d["small_launches"].items() if isinstance(v, dict)}))
for g in r["regimes"]:
    print("  %dx%d%s: %.0f q/s, %.3f ms/step, launch %.3f ms, frac/launch %.3f, over wall %.3f" % (g["batches_per_launch"], g["launches_in_flight"], "" if g["gated"] else "u",
          g["queries_per_s"], g["ms_per_step"], g["avg_kernel_ms"], g["frac_per_launch"], g["frac_over_wall"]))
PY
