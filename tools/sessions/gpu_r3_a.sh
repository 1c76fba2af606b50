#!/bin/bash
# round 3, GPU session A: the whole -m gpu suite (new: configs[3] as 8 co-resident shards), the driver's bench command
# (reference agreement at 10M rows), bench.py --config c4.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider) > $O/r3a_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3a_pytest.txt
(time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5) > $O/r3a_bench_c3.json 2> $O/r3a_bench_c3.err; echo "bench c3 rc $?"; tail -c 600 $O/r3a_bench_c3.err
(time timeout 600 python bench.py --config c4 --steps 32 --warmup 8) > $O/r3a_bench_c4.json 2> $O/r3a_bench_c4.err; echo "bench c4 rc $?"; tail -c 600 $O/r3a_bench_c4.err
python - <<'PY'
import json
for f in ("r3a_bench_c3.json", "r3a_bench_c4.json"):
    try:
        r = json.loads([l for l in open("gpurun_out/" + f) if l.startswith("{")][-1])
        print(f, round(r["value"]), "q/s recall", r["recall_at_10"], "ef", r["ef_search"], "frac", round(r["roofline"]["frac"], 3),
              "build", round(r["build_rows_per_s"]), "agree", (r.get("cpu_baseline") or {}).get("agreement"))
    except Exception as e:
        print(f, "unreadable", e)
PY
