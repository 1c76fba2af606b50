#!/bin/bash
# round 3, GPU session G: batched list insert through ds_bpermute (merge_sorted)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r3g_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3g_pytest.txt
VSS_SEARCH_SOLO=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark and not two_rank and not variants" > $O/r3g_pytest_solo_forced.txt 2>&1; echo "solo-forced pytest rc $?"; tail -n 2 $O/r3g_pytest_solo_forced.txt
timeout 300 python tools/gpu_solo_probe.py 1000000 128 l2sq 16 128 64 > $O/r3g_solo_1m128.txt 2>&1; echo "solo probe rc $?"; grep -E "single|  32:|1024|build" $O/r3g_solo_1m128.txt
VSS_LIBRARY=duckdb-vss_amd/libvssgpu_prof.so timeout 300 python tools/gpu_solo_phase_probe.py 1000000 128 l2sq 16 128 64 > $O/r3g_solo_phase_1m128.txt 2>&1; echo "phase probe rc $?"; cat $O/r3g_solo_phase_1m128.txt
(timeout 300 python bench.py --config c2) > $O/r3g_bench_c2.json 2> $O/r3g_bench_c2.err; echo "bench c2 rc $?"; tail -c 300 $O/r3g_bench_c2.err
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline) > $O/r3g_bench_c3.json 2> $O/r3g_bench_c3.err; echo "bench c3 rc $?"; tail -c 300 $O/r3g_bench_c3.err
python - <<'PY'
import json
try:
    r = json.loads([l for l in open("gpurun_out/r3g_bench_c2.json") if l.startswith("{")][-1])
    print("c2:", round(r["value"]), "q/s", round(r["ms_per_step"] * 1e3, 1), "us/call kernel", round(r["roofline"]["avg_kernel_ms"] * 1e3, 1), "us",
          round(r["roofline"]["us_per_expansion"], 2), "us/expansion; cpu", round(r["cpu_baseline"]["value"]), "build", round(r["build_rows_per_s"]))
    r = json.loads([l for l in open("gpurun_out/r3g_bench_c3.json") if l.startswith("{")][-1])
    print("c3:", round(r["value"]), "q/s frac", round(r["roofline"]["frac"], 3), "build", round(r["build_rows_per_s"]), r["build_kernel_ms"], [(x["batches_per_launch"], x["launches_in_flight"], round(x["queries_per_s"]), round(x["frac_per_launch"], 3)) for x in r["roofline"]["regimes"]])
except Exception as e:
    print("unreadable", e)
PY
