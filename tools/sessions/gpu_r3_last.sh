#!/bin/bash
# round 3, last check of the final tree: whole -m gpu suite, smoke, the driver's bench command
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
rm -f $O/config_tests.txt
(time timeout 1200 python -m pytest tests -q -m gpu -x -p no:cacheprovider) > $O/r3_last_tests.txt 2>&1; echo "pytest rc $?"; tail -n 6 $O/r3_last_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1
(time timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5) > $O/r3_last_bench.json 2> $O/r3_last_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3_last_bench.json") if l.startswith("{")][-1])
r = d["roofline"]
print("value %.0f recall %s ef %d frac %.3f wall %.3f build %.0f traffic %s agree %s cpu %.0f" % (d["value"], d["recall_at_10"], d["ef_search"], r["frac"], r["frac_over_wall"],
      d["build_rows_per_s"], r.get("traffic"), d["cpu_baseline"]["agreement"]["id_match_frac"], d["cpu_baseline"]["value"]))
PY
