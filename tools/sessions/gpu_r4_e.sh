#!/bin/bash
# round 4, GPU session E: the default bench.py run (headline + small launches + extras c2 / c4 / c5 / a13) at full size
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5) > $O/r4e_bench_driver_cmd.json 2> $O/r4e_bench_driver_cmd.err; echo "bench rc $?"; tail -c 600 $O/r4e_bench_driver_cmd.err
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
d = json.loads([l for l in open(O + "/r4e_bench_driver_cmd.json") if l.startswith("{")][-1])
r = d["roofline"]
print("c3: value %.0f q/s, ms/step %.3f, ef %d, recall %.4f +- %.4f (selection %s), frac %.3f over wall %.3f, kernel %.3f ms x %d, build %.0f rows/s" % (
    d["value"], d["ms_per_step"], d["ef_search"], d["recall_at_10"], d["recall_at_10_se"], d["recall"]["selection"], r["frac"], r["frac_over_wall"],
    r["avg_kernel_ms"], r["launches"], d["build_rows_per_s"]))
for g in r["regimes"]:
    print("  %dx%d%s: %.0f q/s, launch %.3f ms, frac/launch %.3f, over wall %.3f" % (g["batches_per_launch"], g["launches_in_flight"],
          "" if g["gated"] else "u", g["queries_per_s"], g["avg_kernel_ms"], g["frac_per_launch"], g["frac_over_wall"]))
print("small launches:", json.dumps(d["small_launches"]))
print("agreement:", json.dumps(d["cpu_baseline"]["agreement"]))
print("cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"] and d["cpu_baseline"]["all_cores"]["search_queries_per_s"])
print("build kernel ms:", d["build_kernel_ms"], "build roofline:", d["build_roofline"]["achieved"])
print("extras:", d.get("extras"))
for name in (d.get("extras") or {}).get("configs", []):
    e = d[name]
    print(name, {k: e.get(k) for k in ("error", "value", "unit", "wall_s", "exit_code", "ef_search", "recall_at_10", "recall_at_100", "build_rows_per_s")},
          "frac", (e.get("roofline") or {}).get("frac"), "agreement", (e.get("cpu_baseline") or {}).get("agreement"),
          "cpu", (e.get("cpu_baseline") or {}).get("value"))
    if name == "c2": print("   join_chunk", e.get("join_chunk"))
    if name == "c5": print("   crud", e.get("crud"))
PY
