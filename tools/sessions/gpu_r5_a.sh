#!/bin/bash
# round 5, session A: (1) the configs[4]-shard line at FULL size (12.5M x 1536, ef 480) with the compact visited sets and with the
# plain ones; (2) the driver's command with round 5's output contract (compact LAST line, extras on earlier lines, sidecars).
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
for c in 1 0; do
  VSS_VISITED_COMPACT=$c timeout 600 python bench.py --config c5 --steps 32 --warmup 16 --cpu-seconds 8 --sidecar $O/r5a_c5_compact$c.full.json > $O/r5a_bench_c5_compact$c.json 2> $O/r5a_bench_c5_compact$c.err; echo "c5 compact=$c rc $?"
  tail -c 1500 $O/r5a_bench_c5_compact$c.json
done
(time timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 --sidecar $O/r5a_driver_cmd.full.json > $O/r5a_bench_driver_cmd.json 2> $O/r5a_bench_driver_cmd.err); echo "driver cmd rc $?"
tail -c 8000 $O/r5a_bench_driver_cmd.json > $O/r5a_driver_tail_8000.txt
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
tail = open(O + "/r5a_driver_tail_8000.txt").read()
last = [l for l in tail.splitlines() if l.startswith("{")][-1]
d = json.loads(last)
print("LAST LINE %d chars; parsed keys: %s" % (len(last), sorted(d)))
print("headline %.0f q/s recall %.4f frac %.3f cpu %s" % (d["value"], d["recall_at_10"], d["roofline"]["frac"], d["cpu_baseline"]["value"]))
for l in tail.splitlines():
    if l.startswith('{"extra"'):
        print(l[:400])
PY
