#!/bin/bash
# round 3, GPU session J: solo shape — accept phase in the shadow of the next expansion's row loads
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r3j_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3j_pytest.txt
VSS_SEARCH_SOLO=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark and not two_rank and not variants" > $O/r3j_pytest_solo_forced.txt 2>&1; echo "solo-forced pytest rc $?"; tail -n 2 $O/r3j_pytest_solo_forced.txt
VSS_SEARCH_OVERLAP=0 timeout 300 python tools/gpu_solo_probe.py 1000000 128 l2sq 16 128 64 > $O/r3j_solo_overlap_off.txt 2>&1; echo "overlap off:"; grep -E "solo" $O/r3j_solo_overlap_off.txt | grep -E "single|  32:|1024"
timeout 300 python tools/gpu_solo_probe.py 1000000 128 l2sq 16 128 64 > $O/r3j_solo_overlap_on.txt 2>&1; echo "overlap on:"; grep -E "single|  32:|1024|build" $O/r3j_solo_overlap_on.txt
(timeout 300 python bench.py --config c2) > $O/r3j_bench_c2.json 2> $O/r3j_bench_c2.err; echo "bench c2 rc $?"
python - <<'PY'
import json
try:
    r = json.loads([l for l in open("gpurun_out/r3j_bench_c2.json") if l.startswith("{")][-1])
    print("c2:", round(r["value"]), "q/s", round(r["ms_per_step"] * 1e3, 1), "us/call kernel", round(r["roofline"]["avg_kernel_ms"] * 1e3, 1), "us",
          round(r["roofline"]["us_per_expansion"], 2), "us/expansion; cpu", round(r["cpu_baseline"]["value"]), "ratio", round(r["value"] / r["cpu_baseline"]["value"], 3), r["cpu_baseline"]["agreement"]["id_match_frac"])
except Exception as e:
    print("unreadable", e)
PY
