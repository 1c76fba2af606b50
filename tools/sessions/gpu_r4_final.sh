#!/bin/bash
# round 4 evidence: whole -m gpu suite, smoke, the driver's bench command (plain with all extras; under rocprofv3 --kernel-trace;
# under the two PMC passes on the same launch shape), row a13 under rocprofv3 (trace + FETCH_SIZE), the exact path folded / plain
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
P=$O/prof_r04
mkdir -p $P/summary
cd $R
export TMPDIR=/tmp
rm -f $O/config_tests.txt
if [ "$1" != "nosuite" ]; then
  (time timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 -p no:cacheprovider) > $O/r4_final_tests.txt 2>&1; echo "pytest rc $?"
  tail -n 14 $O/r4_final_tests.txt
  cat $O/config_tests.txt 2>/dev/null
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r4_final_smoke.txt 2>&1; tail -n 2 $O/r4_final_smoke.txt
fi
(time timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5) > $P/bench_plain.json 2> $P/bench_plain.err; echo "driver-style bench rc $?"; tail -n 4 $P/bench_plain.err
EF=$(python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r04"
print(json.loads([l for l in open(O + "/bench_plain.json") if l.startswith("{")][-1])["ef_search"])
PY
)
echo "ef_search chosen by the rule: $EF"
BARE="--gpus 1 --steps 20 --warmup 5 --ef $EF --regimes none --no-cpu-baseline --host-api-seconds 0 --extras none --no-small-launches --heldout-batches 0"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $P/kt -o bench -- python3 $R/bench.py $BARE > $P/bench_under_rocprof.json 2> $P/kt.err; echo "rocprof rc $?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex k_search -d $P/pmc_$c -o pmc -- python3 $R/bench.py $BARE > $P/bench_pmc_$c.json 2> $P/pmc_$c.err; echo "pmc $c rc $?"
done
timeout 200 rocprofv3 --kernel-trace --stats -d $P/a13_kt -o a13 -- python3 $R/bench.py --config a13 --steps 20 > $P/a13_under_rocprof.json 2> $P/a13_kt.err; echo "a13 rocprof rc $?"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex k_array_distance -d $P/a13_pmc -o pmc -- python3 $R/bench.py --config a13 --steps 20 > $P/a13_pmc.json 2> $P/a13_pmc.err; echo "a13 pmc rc $?"
cd $R && python - <<'PY'
import csv, json, os, sqlite3
P = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r04"
S = P + "/summary"
def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
db = sqlite3.connect(P + "/kt/bench_results.db")
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
with open(S + "/r04_bench_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, mn, mx in rows:
        w.writerow([name[:110], calls, tot, "%.1f" % avg, "%.4f" % (100.0 * tot / total), mn, mx])
ks = db.execute("select name, start, end from kernels where name like '%k_search%' order by start").fetchall()
t0 = ks[0][1]
with open(S + "/r04_k_search_trace.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Launch", "StartNs", "EndNs", "DurationNs", "GapSincePreviousEndNs", "Name"])
    prev = None
    for i, (name, st, en) in enumerate(ks):
        w.writerow([i, st - t0, en - t0, en - st, "" if prev is None else st - prev, name[:60]])
        prev = en
under = last_json(P + "/bench_under_rocprof.json")
json.dump(under, open(S + "/r04_bench_under_rocprof.json", "w"), indent=1)
json.dump(last_json(P + "/bench_plain.json"), open(S + "/r04_bench_latest.json", "w"), indent=1)
timed = sorted((en - st for _, st, en in ks), reverse=True)[:under["roofline"]["launches"]]
print("rocprof: timed k_search launches", [round(t / 1e6, 3) for t in timed], "ms; bench.py hipEvents avg", round(under["roofline"]["avg_kernel_ms"], 3), "ms")
out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = sqlite3.connect(P + "/pmc_%s/pmc_results.db" % counter)
    out[counter] = [r[0] for r in d.execute("select sum(value) from counters_collection where counter_name = ? and kernel_name like '%k_search%' group by dispatch_id order by dispatch_id", (counter,)).fetchall()]
cfg = last_json(P + "/bench_pmc_FETCH_SIZE.json")
n_timed = cfg["roofline"]["launches"]
per_launch = cfg["steps"] / n_timed
top = sorted(range(len(out["FETCH_SIZE"])), key=lambda i: -out["FETCH_SIZE"][i])[:n_timed]
fetch = sum(out["FETCH_SIZE"][i] for i in top) / n_timed
write = sum(sorted(out["WRITE_SIZE"], reverse=True)[:n_timed]) / n_timed
summary = {
    "kernel": "k_search<1, 3, 4, 2> (crews + pipelined level search), the timed launches of the driver's command: %d launches of %g batches x 1024 queries (vss_search_multi_device_begin)" % (n_timed, per_launch),
    "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-include-regex k_search -- python3 bench.py --gpus 1 --steps 20 --warmup 5 "
               "--ef %d --regimes none --no-cpu-baseline --host-api-seconds 0 --extras none --no-small-launches --heldout-batches 0 (two passes)" % cfg["ef_search"],
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
json.dump(summary, open(S + "/r04_pmc_k_search_driver_shape.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "all_launches_FETCH_SIZE"}, indent=1))
# row a13: kernel trace + FETCH_SIZE of k_array_distance
try:
    a = last_json(P + "/a13_under_rocprof.json")
    d = sqlite3.connect(P + "/a13_kt/a13_results.db")
    kr = d.execute("select name, count(*), avg(end-start), min(end-start), max(end-start) from kernels where name like '%k_array_distance%' group by name").fetchall()
    dp = sqlite3.connect(P + "/a13_pmc/pmc_results.db")
    fetch = [r[0] for r in dp.execute("select sum(value) from counters_collection where counter_name = 'FETCH_SIZE' and kernel_name like '%k_array_distance%' group by dispatch_id order by dispatch_id").fetchall()]
    rows_, dim_ = a["config"]["rows"], a["config"]["dim"]
    per_leg = len(fetch) // 6 if fetch else 0
    legs = []
    for i, leg in enumerate(a["legs"]):
        vals = fetch[i * per_leg:(i + 1) * per_leg] if per_leg else []
        legs.append(dict(leg, FETCH_SIZE_mean_KiB=(sum(vals) / len(vals)) if vals else None,
                         hbm_read_bytes=(sum(vals) / len(vals) * 1024 * 2) if vals else None))
    json.dump({"command": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE --kernel-include-regex k_array_distance -- python3 bench.py --config a13 --steps 20",
               "line": a, "kernel_trace": [{"name": n[:100], "dispatches": c, "avg_ns": av, "min_ns": mn, "max_ns": mx} for n, c, av, mn, mx in kr],
               "legs_with_traffic": legs,
               "corrections": "FETCH_SIZE in KiB, doubled for 16-B/lane coalesced reads on gfx950 (MI355X_MICROARCH.md)",
               "parity": "UNPINNED (DuckDB v1.4.3 core source absent)"}, open(S + "/r04_a13_array_functions_rocprof.json", "w"), indent=1)
    print("a13:", [(l["function"], l["operand"], round(l["frac"], 3)) for l in a["legs"]], "trace", kr)
except Exception as e:
    print("a13 summary failed:", repr(e))
PY
rm -rf $P/kt $P/pmc_FETCH_SIZE $P/pmc_WRITE_SIZE $P/a13_kt $P/a13_pmc
cd $R
timeout 300 python tools/gpu_exact_filter_probe.py 4000000 2>&1 | grep -v amdgpu | tee $O/r4_final_exact_filter_probe.txt
if [ "$2" == "footprint" ]; then
  timeout 500 python tools/gpu_footprint_probe.py 768 10000000 25000000 2>&1 | grep -v amdgpu | tee $O/r4_final_footprint_probe.txt
fi
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for f in ("prof_r04/bench_plain.json", "prof_r04/bench_under_rocprof.json"):
    try:
        d = json.loads([l for l in open(O + "/" + f).read().splitlines() if l.startswith("{")][-1])
        r = d["roofline"]
        print("%s: %gx%d steps %d ef %d value %.0f ms/step %.3f frac/launch %.3f kernel ms %.3f (%d launches) over wall %.3f recall %s+-%s build %.0f traffic %s" % (
            f, d["config"]["batches_per_launch_timed"], d["config"]["launches_in_flight"], d["steps"], d["ef_search"], d["value"], d["ms_per_step"], r["frac"],
            r["avg_kernel_ms"], r["launches"], r["frac_over_wall"], d["recall_at_10"], d.get("recall_at_10_se"), d["build_rows_per_s"], r.get("traffic")))
        for g in r["regimes"]:
            print("  %dx%d%s: %.0f q/s, launch %.3f ms, frac/launch %.3f, over wall %.3f" % (
                g["batches_per_launch"], g["launches_in_flight"], "" if g["gated"] else "u", g["queries_per_s"], g["avg_kernel_ms"], g["frac_per_launch"], g["frac_over_wall"]))
        if d.get("small_launches"):
            print("  small launches:", d["small_launches"]["single_query"], d["small_launches"]["join_chunk"])
        if d.get("cpu_baseline"):
            print("  cpu:", d["cpu_baseline"]["value"], "agreement", d["cpu_baseline"]["agreement"])
        print("  exact batch s", d.get("exact_batch_s"), "host api", d.get("host_api_queries_per_s"), "extras", d.get("extras"))
        for name in (d.get("extras") or {}).get("configs", []):
            e = d[name]
            a = (e.get("cpu_baseline") or {}).get("agreement") or {}
            print("  ", name, {k: e.get(k) for k in ("error", "value", "unit", "wall_s", "exit_code", "ef_search", "recall_at_10", "recall_at_100", "build_rows_per_s")},
                  "frac", (e.get("roofline") or {}).get("frac"), "agreement ids", a.get("id_match_frac"), "unexplained", a.get("unexplained_mismatches"),
                  "cpu", (e.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "unreadable:", repr(e))
PY
