#!/bin/bash
# HBM traffic counters of k_search on the launch shape of the timed region (8 batches per launch)
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/gpurun_out/prof_r02c
mkdir -p $P/summary
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex k_search -d $P/pmc8_$c -o pmc -- python3 $R/bench.py --steps 32 --warmup 8 --ef 96 --regimes none --no-cpu-baseline --host-api-seconds 0 > $P/bench_pmc8_$c.json 2> $P/pmc8_$c.err; echo "pmc $c rc $?"
done
cd $R && python - <<'PY'
import json, os, sqlite3
P = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r02c"
out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = sqlite3.connect(P + "/pmc8_%s/pmc_results.db" % counter)
    cols = [r[1] for r in d.execute("pragma table_info(counters_collection)")]
    rows = d.execute("select value from counters_collection where counter_name = ? and kernel_name like '%k_search%' "
                     "order by value", (counter,)).fetchall()
    vals = [r[0] for r in rows]
    out[counter + "_all_launches"] = vals
    if counter == "FETCH_SIZE":
        top = max(vals)
        full = [v for v in vals if v > 0.9 * top]  # the launches that carried 8 batches
        out["full_launches"] = len(full)
        out["FETCH_SIZE_mean"] = sum(full) / len(full)
        n_full = len(full)
    else:
        vals_desc = sorted(vals, reverse=True)[:n_full]
        out["WRITE_SIZE_mean"] = sum(vals_desc) / len(vals_desc)
cfg = json.loads([l for l in open(P + "/bench_pmc8_FETCH_SIZE.json").read().splitlines() if l.startswith("{")][-1])
summary = {
    "kernel": "k_search<1, 3, 4, 2>, launches of 8 batches x 1024 queries (vss_search_multi_device_begin)",
    "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-include-regex k_search -- python3 bench.py "
               "--steps 32 --warmup 8 --ef 96 --regimes none --no-cpu-baseline --host-api-seconds 0 (two passes)",
    "config": {k: cfg["config"][k] for k in ("rows", "dim", "index_metric", "M", "M0", "ef_construction", "ef_search",
                                             "batch_queries", "k")},
    "batches_per_launch": 8, "launches": out["full_launches"],
    "FETCH_SIZE_mean": round(out["FETCH_SIZE_mean"], 2), "WRITE_SIZE_mean": round(out["WRITE_SIZE_mean"], 2),
    "corrections": "bytes = counter * 1024; FETCH_SIZE doubled for 16-B/lane coalesced reads on gfx950 "
                   "(MI355X_MICROARCH.md, HBM section); only the launches that carried 8 batches are averaged",
    "hbm_bytes_per_launch": out["FETCH_SIZE_mean"] * 1024 * 2 + out["WRITE_SIZE_mean"] * 1024,
    "algorithmic_bytes_per_launch_in_that_run": cfg["roofline"]["algorithmic_bytes_per_launch"],
    "all_launches_FETCH_SIZE": out["FETCH_SIZE_all_launches"],
}
json.dump(summary, open(P + "/summary/r02c_pmc_k_search_8_batches.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "all_launches_FETCH_SIZE"}, indent=1))
PY
rm -rf $P/pmc8_FETCH_SIZE $P/pmc8_WRITE_SIZE
