#!/bin/bash
# round 3, GPU session W: descent with the neighbours' upper-list offsets fetched in the shadow of their rows — parity, A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r3w_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3w_pytest.txt
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --config c2 --steps 4000 --no-cpu-baseline > $O/r3w_c2_$name.json 2> $O/r3w_c2_$name.err; echo "c2 $name rc $?"; }
run ahead A=1
run plain VSS_DESCENT_AHEAD=0
run ahead_again A=1
run plain_again VSS_DESCENT_AHEAD=0
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3w_c2_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "q/s %.0f" % r["value"], "us/call %.1f" % (r["ms_per_step"] * 1e3), "kernel us %.1f" % (r["roofline"]["avg_kernel_ms"] * 1e3), "build rows/s %.0f" % r["build_rows_per_s"])
    except Exception as e:
        print(f, "unreadable", e)
PY
for v in 1 0; do echo "VSS_DESCENT_AHEAD=$v" | tee -a $O/r3w_dim768.txt; VSS_DESCENT_AHEAD=$v timeout 300 python tools/gpu_dim_probe.py 2000000 768 2>&1 | grep -v amdgpu | tee -a $O/r3w_dim768.txt; done
