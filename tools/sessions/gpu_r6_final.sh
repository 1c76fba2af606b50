#!/bin/bash
# round 6 evidence: [whole -m gpu suite, smoke,] the driver's bench command (plain, all extras; its last 8 000 characters kept as the
# driver sees them), the same command under rocprofv3 --kernel-trace, and under the two PMC passes on the same launch shape.
#   bash tools/sessions/gpu_r6_final.sh [nosuite]
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
P=$O/prof_r06
mkdir -p $P/summary
cd $R
export TMPDIR=/tmp
if [ "$1" != "nosuite" ]; then
  (time timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 -p no:cacheprovider) > $O/r6_final_tests.txt 2>&1; echo "pytest rc $?"
  tail -n 14 $O/r6_final_tests.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r6_final_smoke.txt 2>&1; tail -n 2 $O/r6_final_smoke.txt
fi
(time timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 --sidecar $P/summary/r06_bench_latest_sidecar.json) > $P/bench_plain.jsonl 2> $P/bench_plain.err; echo "driver-style bench rc $?"; tail -n 4 $P/bench_plain.err
tail -c 8000 $P/bench_plain.jsonl > $P/summary/r06_bench_latest_last_8000_chars.txt
cp $P/bench_plain.jsonl $P/summary/r06_bench_latest_stdout.jsonl
EF=$(python - <<'PY'
import json, os
P = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r06"
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
P = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r06"
S = P + "/summary"
def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
def db(d):
    return sqlite3.connect(sorted(glob.glob(P + "/" + d + "/**/*.db", recursive=True))[0])
d = db("kt")
rows = d.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
with open(S + "/r06_bench_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, mn, mx in rows:
        w.writerow([name[:110], calls, tot, "%.1f" % avg, "%.4f" % (100.0 * tot / total), mn, mx])
ks = d.execute("select name, start, end from kernels where name like '%k_search%' order by start").fetchall()
t0 = ks[0][1]
with open(S + "/r06_k_search_trace.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Launch", "StartNs", "EndNs", "DurationNs", "GapSincePreviousEndNs", "Name"])
    prev = None
    for i, (name, st, en) in enumerate(ks):
        w.writerow([i, st - t0, en - t0, en - st, "" if prev is None else st - prev, name[:60]])
        prev = en
under = last_json(P + "/bench_under_rocprof.jsonl")
json.dump(under, open(S + "/r06_bench_under_rocprof.json", "w"), indent=1)
json.dump(last_json(P + "/bench_plain.jsonl"), open(S + "/r06_bench_latest.json", "w"), indent=1)
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
    "kernel": "k_search<1, 3, 4, 2, 1024> (crews + pipelined level search, blocked candidate list), the timed launches of the driver's command: %d launches of %g batches x 1024 queries (vss_search_multi_device_begin)" % (n_timed, per_launch),
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
json.dump(summary, open(S + "/r06_pmc_k_search_driver_shape.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "all_launches_FETCH_SIZE"}, indent=1))
PY
rm -rf $P/kt $P/pmc_FETCH_SIZE $P/pmc_WRITE_SIZE
python - <<'PY'
import json, os
P = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r06"
tail = open(P + "/summary/r06_bench_latest_last_8000_chars.txt").read()
last = [l for l in tail.splitlines() if l.startswith("{")][-1]
d = json.loads(last)
print("LAST LINE %d chars" % len(last))
print("headline %.0f q/s recall %.4f frac %.3f traffic/alg %s cpu %s build %.0f rows/s" % (d["value"], d["recall_at_10"], d["roofline"]["frac"], d["roofline"].get("traffic_over_algorithmic"), d["cpu_baseline"]["value"], d["build_rows_per_s"]))
for l in open(P + "/bench_plain.jsonl").read().splitlines():
    if l.startswith('{"extra"') or l.startswith('{"detail": "exact"') or l.startswith('{"detail": "small') or l.startswith('{"detail": "regime'):
        print(l[:420])
PY
# ---- the configs[4] shard at full size under the counters, 32 batches per launch (the launch shape of bench.py --config c5)
if [ "$1" == "c5pmc" ] || [ "$2" == "c5pmc" ]; then
P5=$O/prof_r06_c5
mkdir -p $P5
BARE5="--config c5 --steps 32 --warmup 16 --ef 480 --no-cpu-baseline"
cd /tmp
timeout 500 rocprofv3 --kernel-trace -d $P5/kt -o c5 -- python3 $R/bench.py $BARE5 --sidecar $P5/kt_full.json > $P5/c5_under_rocprof.jsonl 2> $P5/kt.err; echo "c5 rocprof rc $?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex k_search -d $P5/pmc_$c -o pmc -- python3 $R/bench.py $BARE5 --sidecar $P5/pmc_${c}_full.json > $P5/c5_pmc_$c.jsonl 2> $P5/pmc_$c.err; echo "c5 pmc $c rc $?"
done
cd $R && python - <<'PY'
import glob, json, os, sqlite3
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
P = R + "/gpurun_out/prof_r06_c5"
def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
def db(d):
    return sqlite3.connect(sorted(glob.glob(P + "/" + d + "/**/*.db", recursive=True))[0])
under = last_json(P + "/c5_under_rocprof.jsonl")
ks = db("kt").execute("select name, start, end from kernels where name like '%k_search%' order by start").fetchall()
big = sorted(((en - st, name) for name, st, en in ks), reverse=True)[:3]
out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    out[counter] = [r[0] for r in db("pmc_" + counter).execute("select sum(value) from counters_collection where counter_name = ? and kernel_name like '%k_search%' group by dispatch_id order by dispatch_id", (counter,)).fetchall()]
cfg = last_json(P + "/c5_pmc_FETCH_SIZE.jsonl")
n_timed = cfg["roofline"]["launches"]
fetch = sum(sorted(out["FETCH_SIZE"], reverse=True)[:n_timed]) / n_timed
write = sum(sorted(out["WRITE_SIZE"], reverse=True)[:n_timed]) / n_timed
summary = {
    "kernel": "k_search<2, 6, 4, 8, 768> (12 waves, pipelined level search, blocked 8-register list, compact visited sets that MOVE to HBM when they outgrow LDS): the timed launches of bench.py --config c5 — %d launch(es) of %g batches x 1024 queries, top-100, ef 480, one 12.5M x 1536 ip shard" % (n_timed, cfg["steps"] / n_timed),
    "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-include-regex k_search -- python3 bench.py --config c5 --steps 32 --warmup 16 --ef 480 --no-cpu-baseline (two passes; a third with --kernel-trace only)",
    "config": {k: cfg["config"][k] for k in ("rows", "dim", "index_metric", "M", "M0", "ef_construction", "ef_search", "batch_queries", "k")},
    "batches_per_launch": cfg["steps"] / n_timed, "launches": n_timed, "FETCH_SIZE_mean": round(fetch, 2), "WRITE_SIZE_mean": round(write, 2),
    "corrections": "bytes = counter * 1024; FETCH_SIZE doubled for 16-B/lane coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section); the timed launches = the dispatches with the largest counter values",
    "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
    "algorithmic_bytes_per_launch_in_that_run": cfg["roofline"]["algorithmic_bytes_per_launch"],
    "frac_in_that_run": cfg["roofline"]["frac"], "frac_under_kernel_trace_only": under["roofline"]["frac"],
    "longest_k_search_launches_ms_kernel_trace": [round(t / 1e6, 3) for t, _ in big],
    "all_launches_FETCH_SIZE": out["FETCH_SIZE"],
}
summary["traffic_over_algorithmic"] = summary["hbm_bytes_per_launch"] / summary["algorithmic_bytes_per_launch_in_that_run"]
json.dump(summary, open(R + "/gpurun_out/r06_pmc_k_search_config4_shard_full_size.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "all_launches_FETCH_SIZE"}, indent=1))
PY
rm -rf $P5/kt $P5/pmc_FETCH_SIZE $P5/pmc_WRITE_SIZE
fi
# ---- the bulk build at the headline's options (M 32, ef_construction 384) under the counters (the only pass was
# round 4's, at ef_construction 256)
if [ "$1" == "buildpmc" ] || [ "$2" == "buildpmc" ] || [ "$3" == "buildpmc" ]; then
PB=$O/build_pmc_r06
mkdir -p $PB
cd /tmp
for pass in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 500 rocprofv3 --kernel-trace --pmc $pass --kernel-include-regex "k_build_phase" -d $PB/pmc_$tag -o pmc -- python $R/tools/gpu_build_probe.py 10000000 0 384 > $PB/probe_$tag.txt 2> $PB/pmc_$tag.err; echo "build pmc $tag rc $?"
done
cd $R && python - "$PB" <<'PY'
import glob, json, os, sqlite3, sys
out = sys.argv[1]
res = {"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum} --kernel-include-regex k_build_phase -- "
                  "python tools/gpu_build_probe.py 10000000 0 384   (three passes; the driver shape: 10M x 768 cosine, M 32, ef_construction 384)"}
for line in open(os.path.join(out, "probe_FETCH_SIZE.txt")):
    if line.startswith("{"):
        res["probe"] = json.loads(line)
for tag, counters in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]), ("TCC_HIT_sum", ["TCC_HIT_sum", "TCC_MISS_sum"])):
    try:
        d = sqlite3.connect(sorted(glob.glob(os.path.join(out, "pmc_%s" % tag, "**", "*.db"), recursive=True))[0])
        for c in counters:
            for name, n, total in d.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                                            "group by kernel_name", (c,)):
                key = "phase_a" if "phase_a" in name else "phase_b"
                res.setdefault(key, {})["kernel"] = name[:100]
                res[key]["launches"] = n
                res[key][c] = total
    except Exception as e:
        res["error_" + tag] = repr(e)
p = res.get("probe", {})
for key in ("phase_a", "phase_b"):
    k = res.get(key)
    if not k:
        continue
    k["hbm_bytes"] = k.get("FETCH_SIZE", 0) * 1024 * 2 + k.get("WRITE_SIZE", 0) * 1024
    if k.get("TCC_HIT_sum") is not None and k.get("TCC_MISS_sum"):
        k["l2_hit_rate"] = k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"])
if p and "phase_a" in res:
    res["phase_a"]["algorithmic_bytes"] = p["phase_a_algorithmic_bytes"]
    res["phase_a"]["fetched_over_algorithmic"] = res["phase_a"]["hbm_bytes"] / p["phase_a_algorithmic_bytes"]
if p and "phase_b" in res:
    alg_b = p["phase_b_distances"] * (4 * p["dim"] + 4)
    res["phase_b"]["algorithmic_bytes"] = alg_b
    res["phase_b"]["fetched_over_algorithmic"] = res["phase_b"]["hbm_bytes"] / alg_b
res["corrections"] = "bytes = counter * 1024; FETCH_SIZE doubled for 16-B/lane coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section)"
json.dump(res, open(os.path.join(os.path.dirname(out), "r06_pmc_build_10m768_efc384.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
rm -rf $PB/pmc_*
fi
