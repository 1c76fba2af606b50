#!/bin/bash
# round 3, GPU session B: the solo search shape and the reordering compaction — parity first, then what they buy.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "compact or variants or both_engine_shapes or tombstones or reference_sql or harness or several_batches or pipelined or limits_beyond or fuzz") > $O/r3b_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3b_pytest.txt
timeout 300 python tools/gpu_solo_probe.py 1000000 128 l2sq 16 128 64 > $O/r3b_solo_1m128.txt 2>&1; echo "solo probe rc $?"; cat $O/r3b_solo_1m128.txt
(timeout 300 python bench.py --config c2) > $O/r3b_bench_c2.json 2> $O/r3b_bench_c2.err; echo "bench c2 rc $?"; tail -c 300 $O/r3b_bench_c2.err
python - <<'PY'
import json
try:
    r = json.loads([l for l in open("gpurun_out/r3b_bench_c2.json") if l.startswith("{")][-1])
    print("c2:", round(r["value"]), "q/s", round(r["ms_per_step"] * 1e3, 1), "us/call kernel", round(r["roofline"]["avg_kernel_ms"] * 1e3, 1), "us",
          round(r["roofline"]["us_per_expansion"], 2), "us/expansion; cpu", round(r["cpu_baseline"]["value"]), "agree", r["cpu_baseline"]["agreement"])
except Exception as e:
    print("c2 unreadable", e)
PY
timeout 600 python tools/gpu_locality_probe.py > $O/r3b_locality_10m768.txt 2>&1; echo "locality rc $?"; cat $O/r3b_locality_10m768.txt
