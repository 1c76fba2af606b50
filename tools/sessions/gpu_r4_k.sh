#!/bin/bash
# round 4, GPU session K: visited-set sizing policy for searches beyond limit 128 — parity subset, the configs[4]-shard line
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "variants or wide_lists or limits_beyond or both_engine_shapes or option_space_fuzz or register_queue or several_batches or tombstones or config4") > $O/r4k_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 5 $O/r4k_pytest.txt
timeout 600 python bench.py --config c5 --steps 32 --warmup 16 --cpu-seconds 8 > $O/r4k_bench_c5.json 2> $O/r4k_bench_c5.err; echo "c5 rc $?"
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
d = json.loads([l for l in open(O + "/r4k_bench_c5.json") if l.startswith("{")][-1])
print("c5: value %.0f q/s ef %d recall %.4f+-%.4f frac %.3f dists/q %.0f build %.0f rows/s; crud %s" % (d["value"], d["ef_search"], d["recall_at_100"], d["recall_at_100_se"],
      d["roofline"]["frac"], d["roofline"]["distances_per_query"], d["build_rows_per_s"], [(c["recall_at_100"], round(c["queries_per_s"])) for c in d["crud"]]))
a = d["cpu_baseline"]["agreement"]
print("   agreement ids %.5f cells %d beyond near-tie %d replayed %d identical %d unexplained %d" % (a["id_match_frac"], a["mismatching_cells"], a["cells_beyond_a_near_tie"],
      a["queries_replayed_in_wave_order"], a["queries_replay_identical_to_engine"], a["unexplained_mismatches"]))
PY
