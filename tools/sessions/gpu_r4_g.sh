#!/bin/bash
# round 4, GPU session G: parity subset on the final kernels, configs[1] through the team shape against the workgroup engine
# (crews + pipelined), the headline run with the round's index options (ef_construction 384)
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "variants or both_engine_shapes or sql or host_harness or rccl or sharded_index or several_batches or reference_built") > $O/r4g_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 5 $O/r4g_pytest.txt
for solo in 1 0; do
  VSS_SEARCH_SOLO=$solo timeout 300 python bench.py --config c2 --steps 3000 --cpu-seconds 3 > $O/r4g_bench_c2_solo$solo.json 2> $O/r4g_bench_c2_solo$solo.err; echo "c2 solo=$solo rc $?"
done
(time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extras none) > $O/r4g_bench_c3_efc384.json 2> $O/r4g_bench_c3_efc384.err; echo "c3 rc $?"; tail -c 400 $O/r4g_bench_c3_efc384.err
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for solo in (1, 0):
    d = json.loads([l for l in open(O + "/r4g_bench_c2_solo%d.json" % solo) if l.startswith("{")][-1])
    print("c2 VSS_SEARCH_SOLO=%d: %.0f q/s, %.1f us/call, kernel %.1f us, %.2f us/expansion, join chunk %.0f us, agreement ids %s" % (
        solo, d["value"], d["ms_per_step"] * 1e3, d["roofline"]["avg_kernel_ms"] * 1e3, d["roofline"]["us_per_expansion"],
        d["join_chunk"]["us_per_chunk"], d["cpu_baseline"]["agreement"]["id_match_frac"]))
d = json.loads([l for l in open(O + "/r4g_bench_c3_efc384.json") if l.startswith("{")][-1])
r = d["roofline"]
print("c3: value %.0f q/s, ms/step %.3f, ef %d, recall %.4f +- %.4f (selection %s), frac %.3f over wall %.3f, kernel %.3f ms x %d, build %.0f rows/s (A %.1f s B %.1f s)" % (
    d["value"], d["ms_per_step"], d["ef_search"], d["recall_at_10"], d["recall_at_10_se"], d["recall"]["selection"], r["frac"], r["frac_over_wall"],
    r["avg_kernel_ms"], r["launches"], d["build_rows_per_s"], d["build_kernel_ms"]["phase_a"] / 1e3, d["build_kernel_ms"]["phase_b"] / 1e3))
print("   distances/query %.0f expansions/query %.1f bytes/launch %.3g" % (r["distances_per_query"], r["expansions_per_query"], r["algorithmic_bytes_per_launch"]))
for g in r["regimes"]:
    print("  %dx%d%s: %.0f q/s, launch %.3f ms, frac/launch %.3f, over wall %.3f" % (g["batches_per_launch"], g["launches_in_flight"],
          "" if g["gated"] else "u", g["queries_per_s"], g["avg_kernel_ms"], g["frac_per_launch"], g["frac_over_wall"]))
print("small launches:", json.dumps(d["small_launches"]))
print("agreement:", json.dumps(d["cpu_baseline"]["agreement"]))
print("cpu:", d["cpu_baseline"]["value"], "host api", d["host_api_queries_per_s"], "exact batch s", d["exact_batch_s"])
PY
