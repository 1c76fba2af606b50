#!/bin/bash
# round 3 evidence: whole -m gpu suite, smoke, the driver's bench command (plain, under rocprofv3 --kernel-trace, under the two
# PMC passes on the same launch shape), the default bench, configs[1] line
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
P=$O/prof_r03
mkdir -p $P/summary
cd $R
export TMPDIR=/tmp
rm -f $O/config_tests.txt
if [ "$1" != "nosuite" ]; then
  (time timeout 1200 python -m pytest tests -q -m gpu -x --durations=8 -p no:cacheprovider) > $O/r3_final_tests.txt 2>&1; echo "pytest rc $?"
  tail -n 14 $O/r3_final_tests.txt
  cat $O/config_tests.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r3_final_smoke.txt 2>&1; tail -n 2 $O/r3_final_smoke.txt
fi
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $P/bench_plain.json 2> $P/bench_plain.err; echo "driver-style bench rc $?"
EF=$(python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r03"
print(json.loads([l for l in open(O + "/bench_plain.json") if l.startswith("{")][-1])["ef_search"])
PY
)
echo "ef_search chosen by the sweep: $EF"
timeout 600 python bench.py --no-cpu-baseline --host-api-seconds 0 --regimes 8x1,16x1 > $O/r3_final_bench_default.json 2> $O/r3_final_bench_default.err; echo "default bench rc $?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $P/kt -o bench -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --ef $EF --regimes none --no-cpu-baseline --host-api-seconds 0 > $P/bench_under_rocprof.json 2> $P/kt.err; echo "rocprof rc $?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex k_search -d $P/pmc_$c -o pmc -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --ef $EF --regimes none --no-cpu-baseline --host-api-seconds 0 > $P/bench_pmc_$c.json 2> $P/pmc_$c.err; echo "pmc $c rc $?"
done
cd $R && python - <<'PY'
import csv, json, os, sqlite3
P = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r03"
S = P + "/summary"
def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
# kernel stats + k_search trace of the driver's command
db = sqlite3.connect(P + "/kt/bench_results.db")
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
with open(S + "/r03_bench_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, mn, mx in rows:
        w.writerow([name[:110], calls, tot, "%.1f" % avg, "%.4f" % (100.0 * tot / total), mn, mx])
ks = db.execute("select name, start, end from kernels where name like '%k_search%' order by start").fetchall()
t0 = ks[0][1]
with open(S + "/r03_k_search_trace.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Launch", "StartNs", "EndNs", "DurationNs", "GapSincePreviousEndNs", "Name"])
    prev = None
    for i, (name, st, en) in enumerate(ks):
        w.writerow([i, st - t0, en - t0, en - st, "" if prev is None else st - prev, name[:60]])
        prev = en
under = last_json(P + "/bench_under_rocprof.json")
json.dump(under, open(S + "/r03_bench_under_rocprof.json", "w"), indent=1)
json.dump(last_json(P + "/bench_plain.json"), open(S + "/r03_bench_latest.json", "w"), indent=1)
# the timed launches = the longest k_search dispatches
timed = sorted((en - st for _, st, en in ks), reverse=True)[:under["roofline"]["launches"]]
print("rocprof: timed k_search launches", [round(t / 1e6, 3) for t in timed], "ms; bench.py hipEvents avg", round(under["roofline"]["avg_kernel_ms"], 3), "ms")
out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = sqlite3.connect(P + "/pmc_%s/pmc_results.db" % counter)
    vals = [r[0] for r in d.execute("select sum(value) from counters_collection where counter_name = ? and kernel_name like '%k_search%' group by dispatch_id order by dispatch_id", (counter,)).fetchall()]
    out[counter] = vals
cfg = last_json(P + "/bench_pmc_FETCH_SIZE.json")
n_timed = cfg["roofline"]["launches"]
per_launch = cfg["steps"] / n_timed
top = sorted(range(len(out["FETCH_SIZE"])), key=lambda i: -out["FETCH_SIZE"][i])[:n_timed]
fetch = sum(out["FETCH_SIZE"][i] for i in top) / n_timed
write = sum(sorted(out["WRITE_SIZE"], reverse=True)[:n_timed]) / n_timed
summary = {
    "kernel": "k_search<1, 3, 4, 2>, the timed launches of the driver's command: %d launches of %g batches x 1024 queries (vss_search_multi_device_begin)" % (n_timed, per_launch),
    "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-include-regex k_search -- python3 bench.py --gpus 1 --steps 20 --warmup 5 "
               "--ef %d --regimes none --no-cpu-baseline --host-api-seconds 0 (two passes)" % cfg["ef_search"],
    "config": dict({k: cfg["config"][k] for k in ("rows", "dim", "index_metric", "M", "M0", "ef_construction", "ef_search", "batch_queries", "k")}, shards=1),
    "batches_per_launch": per_launch, "launches": n_timed,
    "FETCH_SIZE_mean": round(fetch, 2), "WRITE_SIZE_mean": round(write, 2),
    "corrections": "bytes = counter * 1024; FETCH_SIZE doubled for 16-B/lane coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section); "
                   "the timed launches = the dispatches with the largest counter values",
    "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
    "algorithmic_bytes_per_launch_in_that_run": cfg["roofline"]["algorithmic_bytes_per_launch"],
    "all_launches_FETCH_SIZE": out["FETCH_SIZE"],
}
summary["traffic_over_algorithmic"] = summary["hbm_bytes_per_launch"] / summary["algorithmic_bytes_per_launch_in_that_run"]
json.dump(summary, open(S + "/r03_pmc_k_search_driver_shape.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "all_launches_FETCH_SIZE"}, indent=1))
PY
rm -rf $P/kt $P/pmc_FETCH_SIZE $P/pmc_WRITE_SIZE
cd $R
timeout 300 python bench.py --config c2 > $O/r3_final_bench_c2.json 2> $O/r3_final_bench_c2.err; echo "bench c2 rc $?"
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for f in ("prof_r03/bench_plain.json", "r3_final_bench_default.json", "prof_r03/bench_under_rocprof.json"):
    try:
        d = json.loads([l for l in open(O + "/" + f).read().splitlines() if l.startswith("{")][-1])
        r = d["roofline"]
        print("%s: %gx%d steps %d ef %d value %.0f ms/step %.3f frac/launch %.3f kernel ms %.3f (%d launches) over wall %.3f recall %s host_api %s build %.0f traffic %s agree %s" % (
            f, d["config"]["batches_per_launch_timed"], d["config"]["launches_in_flight"], d["steps"], d["ef_search"], d["value"], d["ms_per_step"], r["frac"],
            r["avg_kernel_ms"], r["launches"], r["frac_over_wall"], d["recall_at_10"], d.get("host_api_queries_per_s"), d["build_rows_per_s"], r.get("traffic"),
            (d.get("cpu_baseline") or {}).get("agreement")))
        for g in r["regimes"]:
            print("  %dx%d%s: %.0f q/s, %.3f ms/step, launch %.3f ms, frac/launch %.3f, over wall %.3f" % (
                g["batches_per_launch"], g["launches_in_flight"], "" if g["gated"] else "u", g["queries_per_s"], g["ms_per_step"],
                g["avg_kernel_ms"], g["frac_per_launch"], g["frac_over_wall"]))
    except Exception as e:
        print(f, "unreadable:", e)
d = json.loads(open(O + "/r3_final_bench_c2.json").read().strip().splitlines()[-1])
print("c2:", d["value"], d["ms_per_step"], d["roofline"].get("us_per_expansion"), d["cpu_baseline"]["value"], d["cpu_baseline"]["agreement"])
PY
