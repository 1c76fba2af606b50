#!/bin/bash
# round 5 evidence: [whole -m gpu suite, smoke,] the driver's bench command (plain, all extras; its last 8 000 characters kept as the
# driver sees them), the same command under rocprofv3 --kernel-trace, and under the two PMC passes on the same launch shape.
#   bash tools/sessions/gpu_r5_final.sh [nosuite]
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
P=$O/prof_r05
mkdir -p $P/summary
cd $R
export TMPDIR=/tmp
if [ "$1" != "nosuite" ]; then
  (time timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 -p no:cacheprovider) > $O/r5_final_tests.txt 2>&1; echo "pytest rc $?"
  tail -n 14 $O/r5_final_tests.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r5_final_smoke.txt 2>&1; tail -n 2 $O/r5_final_smoke.txt
fi
(time timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 --sidecar $P/summary/r05_bench_latest_sidecar.json) > $P/bench_plain.jsonl 2> $P/bench_plain.err; echo "driver-style bench rc $?"; tail -n 4 $P/bench_plain.err
tail -c 8000 $P/bench_plain.jsonl > $P/summary/r05_bench_latest_last_8000_chars.txt
cp $P/bench_plain.jsonl $P/summary/r05_bench_latest_stdout.jsonl
EF=$(python - <<'PY'
import json, os
P = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r05"
print(json.loads([l for l in open(P + "/bench_plain.jsonl") if l.startswith("{")][-1])["ef_search"])
PY
)
echo "ef_search chosen by the rule: $EF"
BARE="--gpus 1 --steps 20 --warmup 5 --ef $EF --regimes none --no-cpu-baseline --host-api-seconds 0 --extras none --no-small-launches --heldout-batches 0"
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $P/kt -o bench -- python3 $R/bench.py $BARE --sidecar $P/kt_full.json > $P/bench_under_rocprof.jsonl 2> $P/kt.err; echo "rocprof rc $?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex k_search -d $P/pmc_$c -o pmc -- python3 $R/bench.py $BARE --sidecar $P/pmc_${c}_full.json > $P/bench_pmc_$c.jsonl 2> $P/pmc_$c.err; echo "pmc $c rc $?"
done
cd $R && python - <<'PY'
import csv, glob, json, os, sqlite3
P = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r05"
S = P + "/summary"
def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
def db(d):
    return sqlite3.connect(sorted(glob.glob(P + "/" + d + "/**/*.db", recursive=True))[0])
d = db("kt")
rows = d.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
with open(S + "/r05_bench_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, mn, mx in rows:
        w.writerow([name[:110], calls, tot, "%.1f" % avg, "%.4f" % (100.0 * tot / total), mn, mx])
ks = d.execute("select name, start, end from kernels where name like '%k_search%' order by start").fetchall()
t0 = ks[0][1]
with open(S + "/r05_k_search_trace.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Launch", "StartNs", "EndNs", "DurationNs", "GapSincePreviousEndNs", "Name"])
    prev = None
    for i, (name, st, en) in enumerate(ks):
        w.writerow([i, st - t0, en - t0, en - st, "" if prev is None else st - prev, name[:60]])
        prev = en
under = last_json(P + "/bench_under_rocprof.jsonl")
json.dump(under, open(S + "/r05_bench_under_rocprof.json", "w"), indent=1)
json.dump(last_json(P + "/bench_plain.jsonl"), open(S + "/r05_bench_latest.json", "w"), indent=1)
timed = sorted((en - st for _, st, en in ks), reverse=True)[:under["roofline"]["launches"]]
print("rocprof: timed k_search launches", [round(t / 1e6, 3) for t in timed], "ms; bench.py hipEvents avg", round(under["roofline"]["avg_kernel_ms"], 3), "ms")
out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    out[counter] = [r[0] for r in db("pmc_" + counter).execute("select sum(value) from counters_collection where counter_name = ? and kernel_name like '%k_search%' group by dispatch_id order by dispatch_id", (counter,)).fetchall()]
cfg = last_json(P + "/bench_pmc_FETCH_SIZE.jsonl")
n_timed = cfg["roofline"]["launches"]
per_launch = cfg["steps"] / n_timed
top = sorted(range(len(out["FETCH_SIZE"])), key=lambda i: -out["FETCH_SIZE"][i])[:n_timed]
fetch = sum(out["FETCH_SIZE"][i] for i in top) / n_timed
write = sum(sorted(out["WRITE_SIZE"], reverse=True)[:n_timed]) / n_timed
summary = {
    "kernel": "k_search<1, 3, 4, 2, 1024> (crews + pipelined level search), the timed launches of the driver's command: %d launches of %g batches x 1024 queries (vss_search_multi_device_begin)" % (n_timed, per_launch),
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
json.dump(summary, open(S + "/r05_pmc_k_search_driver_shape.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "all_launches_FETCH_SIZE"}, indent=1))
PY
rm -rf $P/kt $P/pmc_FETCH_SIZE $P/pmc_WRITE_SIZE
python - <<'PY'
import json, os
P = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r05"
tail = open(P + "/summary/r05_bench_latest_last_8000_chars.txt").read()
last = [l for l in tail.splitlines() if l.startswith("{")][-1]
d = json.loads(last)
print("LAST LINE %d chars" % len(last))
print("headline %.0f q/s recall %.4f frac %.3f traffic/alg %s cpu %s build %.0f rows/s" % (d["value"], d["recall_at_10"], d["roofline"]["frac"], d["roofline"].get("traffic_over_algorithmic"), d["cpu_baseline"]["value"], d["build_rows_per_s"]))
for l in open(P + "/bench_plain.jsonl").read().splitlines():
    if l.startswith('{"extra"') or l.startswith('{"detail": "exact"') or l.startswith('{"detail": "small') or l.startswith('{"detail": "regime'):
        print(l[:420])
PY
# the configs[4] shard at full size once more under the counters (after the retry in place): see gpu_r5_l.sh
if [ "$2" == "c5pmc" ] || [ "$1" == "c5pmc" ]; then
  sed -e 's/^(time timeout 900 python -m pytest.*$/echo "(tests skipped here)"/' -e 's/^echo "pytest rc.*$//' $R/tools/sessions/gpu_r5_l.sh > /tmp/l.sh
  bash /tmp/l.sh
fi
