"""Turn the rocprofv3 result databases written by tools/sessions/gpu_profile_round.sh into the summaries committed under
profiles/ (not a pytest module).

    python tools/rocprof_summarize.py gpurun_out/prof_r1c r01c [output dir, default profiles/]

tools/sessions/gpu_profile_round.sh runs it on the GPU box itself (the databases are too big to travel back) into
gpurun_out/prof_<tag>/summary/, from where the files are copied to profiles/.
"""
import csv
import json
import os
import sqlite3
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)

db = sqlite3.connect(os.path.join(src, "kt", "bench_results.db"))
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  "from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
with open(os.path.join(out, "%s_bench_kernel_stats.csv" % tag), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, mn, mx in rows:
        w.writerow([name[:110], calls, tot, "%.1f" % avg, "%.4f" % (100.0 * tot / total), mn, mx])


# start / end of every k_search launch (ns since the first one): the evidence of which launches overlap and which do not
ks = db.execute("select name, start, end from kernels where name like '%k_search%' order by start").fetchall()
if ks:
    t0 = ks[0][1]
    with open(os.path.join(out, "%s_k_search_trace.csv" % tag), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Launch", "StartNs", "EndNs", "DurationNs", "GapSincePreviousEndNs", "Name"])
        prev_end = None
        for i, (name, st, en) in enumerate(ks):
            w.writerow([i, st - t0, en - t0, en - st, "" if prev_end is None else st - prev_end, name[:60]])
            prev_end = en


def last_json(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


under = last_json(os.path.join(src, "bench_under_rocprof.json"))
with open(os.path.join(out, "%s_bench_under_rocprof.json" % tag), "w") as f:
    json.dump(under, f, indent=1)
plain = None
if os.path.exists(os.path.join(src, "bench_plain.json")):
    plain = last_json(os.path.join(src, "bench_plain.json"))
    with open(os.path.join(out, "%s_bench_latest.json" % tag[:3]), "w") as f:
        json.dump(plain, f, indent=1)

if os.path.exists(os.path.join(src, "pmc_FETCH_SIZE", "pmc_results.db")):
    pmc = {}
    kernel = None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = sqlite3.connect(os.path.join(src, "pmc_%s" % counter, "pmc_results.db"))
        r = d.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? "
                      "and kernel_name like '%k_search%' group by kernel_name order by 2 desc", (counter,)).fetchall()
        kernel, launches, mean = r[0]
        pmc[counter] = (launches, mean)
    cfg = last_json(os.path.join(src, "bench_pmc_FETCH_SIZE.json"))["config"]
    summary = {
        "kernel": kernel,
        "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-include-regex k_search -- "
                   "python bench.py --steps 8 --pipeline 1 --coalesce 1 --ef 96 --no-cpu-baseline --host-api-seconds 0",
        "config": {k: cfg[k] for k in ("rows", "dim", "index_metric", "M", "M0", "ef_construction", "ef_search",
                                       "batch_queries", "k")},
        "launches": pmc["FETCH_SIZE"][0],
        "FETCH_SIZE_mean": round(pmc["FETCH_SIZE"][1], 2),
        "WRITE_SIZE_mean": round(pmc["WRITE_SIZE"][1], 2),
        "corrections": "bytes = counter * 1024; FETCH_SIZE doubled for 16-B/lane coalesced reads on gfx950 "
                       "(MI355X_MICROARCH.md, HBM section)",
        "hbm_bytes_per_launch": pmc["FETCH_SIZE"][1] * 1024 * 2 + pmc["WRITE_SIZE"][1] * 1024,
    }
    with open(os.path.join(out, "%s_pmc_k_search.json" % tag), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))
if plain:
    print("plain:", plain["value"], plain["roofline"], plain["build_rows_per_s"], plain["cpu_baseline"])
print("under rocprof:", under["value"], under["roofline"]["avg_kernel_ms"])

# ---- the MFMA kernel of the exact path
if os.path.exists(os.path.join(src, "exact_kt", "exact_results.db")):
    d = sqlite3.connect(os.path.join(src, "exact_kt", "exact_results.db"))
    rows = d.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(os.path.join(out, "%s_exact_kernel_stats.csv" % tag), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, calls, tot, avg, mn, mx in rows:
            w.writerow([name[:110], calls, tot, "%.1f" % avg, "%.4f" % (100.0 * tot / total), mn, mx])
    exact = {"command": "rocprofv3 --kernel-trace [--stats | --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE "
                        "--kernel-include-regex k_exact_scores] -- python tools/gpu_exact_probe.py 1000000",
             "workload": "1024 queries x 1000000 rows x FLOAT[768] cosine, 32768-row chunks: k_exact_scores tile "
                         "1024 x 32768 x 768 per launch (the last chunk is shorter)"}
    sc = [r for r in rows if "k_exact_scores" in r[0]]
    if sc:
        # the full chunks: launches within 5 % of the longest
        full = d.execute("select avg(end-start), count(*) from kernels where name like '%k_exact_scores%' and (end-start) > "
                         "0.95 * (select max(end-start) from kernels where name like '%k_exact_scores%')").fetchone()
        flops = 2.0 * 1024 * 32768 * 768
        exact.update({"k_exact_scores_full_chunk_avg_ns": full[0], "full_chunk_launches": full[1],
                      "flops_per_full_chunk": flops, "tflops": flops / full[0] / 1e3, "peak_tflops_f32_matrix": 157.3,
                      "frac_of_peak": flops / full[0] / 1e3 / 157.3})
    pm = os.path.join(src, "exact_pmc", "pmc_results.db")
    if os.path.exists(pm):
        dp = sqlite3.connect(pm)
        for name, n, mean in dp.execute("select counter_name, count(*), avg(value) from counters_collection where "
                                        "kernel_name like '%k_exact_scores%' group by counter_name"):
            exact["pmc_%s_mean" % name] = mean
            exact["pmc_launches"] = n
    with open(os.path.join(out, "%s_exact_mfma.json" % tag), "w") as f:
        json.dump(exact, f, indent=1)
    print(json.dumps(exact, indent=1))
