#!/bin/bash
# round 5, session L: HBM traffic of k_search on the configs[4] shard at FULL size (12.5M x 1536, ef 480, compact visited sets, 12-wave
# pipelined wide lists): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel trace only), the kernel trace itself, and
# the new collected tests of the round (replicated one-rank RCCL, exact over dims 32/64/128).
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
P=$O/prof_r05_c5
mkdir -p $P
cd $R
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "one_rank_rccl or exact_search_over_many or bulk_build_graph or sequential_build") > $O/r5l_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r5l_pytest.txt
BARE="--config c5 --steps 32 --warmup 16 --ef 480 --no-cpu-baseline"
cd /tmp
timeout 500 rocprofv3 --kernel-trace -d $P/kt -o c5 -- python3 $R/bench.py $BARE --sidecar $P/kt_full.json > $P/c5_under_rocprof.jsonl 2> $P/kt.err; echo "rocprof rc $?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex k_search -d $P/pmc_$c -o pmc -- python3 $R/bench.py $BARE --sidecar $P/pmc_${c}_full.json > $P/c5_pmc_$c.jsonl 2> $P/pmc_$c.err; echo "pmc $c rc $?"
done
cd $R && python - <<'PY'
import glob, json, os, sqlite3
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
P = R + "/gpurun_out/prof_r05_c5"
def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
def db(d):
    return sqlite3.connect(sorted(glob.glob(P + "/" + d + "/**/*.db", recursive=True))[0])
under = last_json(P + "/c5_under_rocprof.jsonl")
ks = db("kt").execute("select name, start, end from kernels where name like '%k_search%' order by start").fetchall()
big = sorted(((en - st, name) for name, st, en in ks), reverse=True)[:3]
print("rocprof: three longest k_search launches", [(round(t / 1e6, 3), n[:40]) for t, n in big], "; bench.py hipEvents avg", round(under["roofline"]["avg_kernel_ms"], 3), "ms")
out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    out[counter] = [r[0] for r in db("pmc_" + counter).execute("select sum(value) from counters_collection where counter_name = ? and kernel_name like '%k_search%' group by dispatch_id order by dispatch_id", (counter,)).fetchall()]
cfg = last_json(P + "/c5_pmc_FETCH_SIZE.jsonl")
n_timed = cfg["roofline"]["launches"]
fetch = sum(sorted(out["FETCH_SIZE"], reverse=True)[:n_timed]) / n_timed
write = sum(sorted(out["WRITE_SIZE"], reverse=True)[:n_timed]) / n_timed
summary = {
    "kernel": "k_search<2, 6, 4, 8, 768> (12 waves, pipelined level search, compact visited sets in LDS): the timed launches of bench.py --config c5 — %d launches of 16 batches x 1024 queries, top-100, ef 480, one 12.5M x 1536 ip shard" % n_timed,
    "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-include-regex k_search -- python3 bench.py --config c5 --steps 32 --warmup 16 --ef 480 --no-cpu-baseline (two passes; a third with --kernel-trace only)",
    "config": {k: cfg["config"][k] for k in ("rows", "dim", "index_metric", "M", "M0", "ef_construction", "ef_search", "batch_queries", "k")},
    "launches": n_timed, "FETCH_SIZE_mean": round(fetch, 2), "WRITE_SIZE_mean": round(write, 2),
    "corrections": "bytes = counter * 1024; FETCH_SIZE doubled for 16-B/lane coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section); the timed launches = the dispatches with the largest counter values",
    "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
    "algorithmic_bytes_per_launch_in_that_run": cfg["roofline"]["algorithmic_bytes_per_launch"],
    "frac_in_that_run": cfg["roofline"]["frac"], "frac_under_kernel_trace_only": under["roofline"]["frac"],
    "longest_k_search_launches_ms_kernel_trace": [round(t / 1e6, 3) for t, _ in big],
    "all_launches_FETCH_SIZE": out["FETCH_SIZE"],
}
summary["traffic_over_algorithmic"] = summary["hbm_bytes_per_launch"] / summary["algorithmic_bytes_per_launch_in_that_run"]
json.dump(summary, open(R + "/gpurun_out/r05_pmc_k_search_config4_shard_full_size.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "all_launches_FETCH_SIZE"}, indent=1))
PY
rm -rf $P/kt $P/pmc_FETCH_SIZE $P/pmc_WRITE_SIZE
