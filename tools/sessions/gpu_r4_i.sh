#!/bin/bash
# round 4, GPU session I: the array_* kernel with rows in flight (tests + the a13 line), configs[4]-shard line with the replay-
# backed agreement gate, the same line at a quarter of the footprint (does 1536-dim search depend on the table size?)
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "array or exact or sql") > $O/r4i_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 4 $O/r4i_pytest.txt
for w in 0 1; do VSS_ARRAY_WIDE=$w timeout 120 python bench.py --config a13 --steps 20 > $O/r4i_a13_wide$w.json 2> /dev/null; echo "a13 wide=$w rc $?"; done
timeout 600 python bench.py --config c5 --steps 32 --warmup 16 --cpu-seconds 8 > $O/r4i_bench_c5.json 2> $O/r4i_bench_c5.err; echo "c5 rc $?"; tail -c 300 $O/r4i_bench_c5.err
timeout 300 python bench.py --config c5 --rows 3000000 --steps 32 --warmup 16 --no-cpu-baseline > $O/r4i_bench_c5_3m_rows.json 2> /dev/null; echo "c5 3M rc $?"
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
def last(p):
    return json.loads([l for l in open(O + "/" + p) if l.startswith("{")][-1])
for w in (0, 1):
    d = last("r4i_a13_wide%d.json" % w)
    print("a13 VSS_ARRAY_WIDE=%d:" % w, [(l["function"][6:], l["operand"], round(l["frac"], 3)) for l in d["legs"]], "spot err", d["spot_check_max_rel_err_vs_fp64"])
for f in ("r4i_bench_c5.json", "r4i_bench_c5_3m_rows.json"):
    d = last(f)
    print(f, "value %.0f q/s ef %d recall %.4f+-%.4f frac %.3f dists/q %.0f build %.0f rows/s; crud %s" % (d["value"], d["ef_search"], d["recall_at_100"], d["recall_at_100_se"],
          d["roofline"]["frac"], d["roofline"]["distances_per_query"], d["build_rows_per_s"], [(c["recall_at_100"], round(c["queries_per_s"])) for c in d["crud"]]))
    if d.get("cpu_baseline"):
        print("   agreement", json.dumps(d["cpu_baseline"]["agreement"]))
PY
