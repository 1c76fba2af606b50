#!/bin/bash
# round 5, session R: the driver's command on the round's final tree, once more (bench.py changed after session O: the configs[4]
# extra attaches its counted traffic) — stdout, its last 8 000 characters, the sidecar.
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
export TMPDIR=/tmp
(time timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --sidecar $O/r5r_bench_sidecar.json) > $O/r5r_bench_stdout.jsonl 2> $O/r5r_bench.err; echo "bench rc $?"; tail -n 3 $O/r5r_bench.err
tail -c 8000 $O/r5r_bench_stdout.jsonl > $O/r5r_bench_last_8000_chars.txt
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
tail = open(O + "/r5r_bench_last_8000_chars.txt").read()
last = [l for l in tail.splitlines() if l.startswith("{")][-1]
d = json.loads(last)
print("LAST LINE %d chars; %.0f q/s recall %.4f frac %.3f traffic/alg %s build %.0f" % (len(last), d["value"], d["recall_at_10"], d["roofline"]["frac"], d["roofline"].get("traffic_over_algorithmic"), d["build_rows_per_s"]))
for l in tail.splitlines():
    if l.startswith('{"extra"'):
        print(l[:300])
PY
