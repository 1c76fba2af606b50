#!/bin/bash
# round 3, GPU session I: build batch cap 65536, configs[4] shard JSON line (ef sweep + CPU baseline), configs[3] on one GPU with
# the shards linked concurrently
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
PROBE_RECALL=1 timeout 300 python tools/gpu_build_probe.py 10000000 65536 > $O/r3i_build_64k.json 2> $O/r3i_build_64k.err; cat $O/r3i_build_64k.json
(time timeout 900 python bench.py --config c5 --steps 32 --warmup 16) > $O/r3i_bench_c5.json 2> $O/r3i_bench_c5.err; echo "bench c5 rc $?"; tail -c 400 $O/r3i_bench_c5.err
(time timeout 600 python bench.py --config c4 --steps 32 --warmup 8) > $O/r3i_bench_c4.json 2> $O/r3i_bench_c4.err; echo "bench c4 rc $?"; tail -c 300 $O/r3i_bench_c4.err
python - <<'PY'
import json
for f in ("r3i_bench_c5.json", "r3i_bench_c4.json"):
    try:
        r = json.loads([l for l in open("gpurun_out/" + f) if l.startswith("{")][-1])
        print(f, round(r["value"]), "q/s", "ef", r["ef_search"], r.get("ef_sweep"), "frac", round(r["roofline"]["frac"], 3), "build", round(r["build_rows_per_s"]),
              "crud", r.get("crud"), "cpu", (r.get("cpu_baseline") or {}).get("value"), "agree", (r.get("cpu_baseline") or {}).get("agreement"))
    except Exception as e:
        print(f, "unreadable", e)
PY
