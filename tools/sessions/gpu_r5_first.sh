#!/bin/bash
# round 5, first GPU session (prepared at the end of round 4, when the GPU minutes had run out): what round 4's last change —
# the compact exact visited set, limits 257-512 — still owes at FULL size.
#   1. the whole -m gpu suite on the tree (the cosine / ip cases of test_compact_visited_set_takes_the_oracles_decisions and
#      tests/test_gpu_configs.py have not met the compact set on a GPU yet)
#   2. the configs[4]-shard line at 12.5M x 1536 with the compact sets and with the plain ones (VSS_VISITED_COMPACT=0)
#   3. the driver's command
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider) > $O/r5a_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 4 $O/r5a_pytest.txt
for c in 1 0; do
  VSS_VISITED_COMPACT=$c timeout 600 python bench.py --config c5 --steps 32 --warmup 16 --cpu-seconds 8 > $O/r5a_bench_c5_compact$c.json 2> $O/r5a_bench_c5_compact$c.err; echo "c5 compact=$c rc $?"
done
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r5a_bench_driver_cmd.json 2> $O/r5a_bench_driver_cmd.err; echo "driver cmd rc $?"
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
def last(p):
    return json.loads([l for l in open(O + "/" + p) if l.startswith("{")][-1])
for c in (1, 0):
    d = last("r5a_bench_c5_compact%d.json" % c)
    print("c5 compact=%d: %.0f q/s ef %d recall %.4f frac %.3f dists/q %.0f (%s); crud %s" % (
        c, d["value"], d["ef_search"], d["recall_at_100"], d["roofline"]["frac"], d["roofline"]["distances_per_query"],
        d["roofline"].get("visited_set"), [(x["recall_at_100"], round(x["queries_per_s"])) for x in d["crud"]]))
d = last("r5a_bench_driver_cmd.json")
print("headline %.0f q/s recall %.4f frac %.3f; c5 in the line: %s" % (d["value"], d["recall_at_10"], d["roofline"]["frac"],
      (d.get("c5") or {}).get("roofline", {}).get("frac")))
PY
