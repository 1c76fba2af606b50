#!/bin/bash
# round 3, GPU session V: helpers follow the rows that can still enter the list (second-level touches) — parity, then A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r3v_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3v_pytest.txt
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --config c2 --steps 4000 --no-cpu-baseline > $O/r3v_c2_$name.json 2> $O/r3v_c2_$name.err; echo "c2 $name rc $?"; }
run follow A=1
run follow_nolisttouch VSS_SEARCH_TOUCH_LISTS=0
run nofollow VSS_SEARCH_TOUCH_FOLLOW=0
run follow_again A=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3v_c2_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "q/s %.0f" % r["value"], "us/call %.1f" % (r["ms_per_step"] * 1e3), "kernel us %.1f" % (r["roofline"]["avg_kernel_ms"] * 1e3),
              "stream-wait us/call %.1f" % r["roofline"].get("us_per_call_waiting_on_the_stream", 0))
    except Exception as e:
        print(f, "unreadable", e)
PY
VSS_SEARCH_TOUCH_FOLLOW=1 timeout 300 python tools/gpu_team_probe.py 1000000 128 l2sq 16 128 64 2>&1 | grep -v amdgpu | tee $O/r3v_team_probe_follow.txt
VSS_SEARCH_TOUCH_FOLLOW=0 timeout 300 python tools/gpu_team_probe.py 1000000 128 l2sq 16 128 64 2>&1 | grep -v amdgpu | tee $O/r3v_team_probe_nofollow.txt
