#!/bin/bash
# round-2 session 11: first-occurrence visits (fuzz seed 10), dim-1536 search kernel with 2 vs 4 rows in flight,
# the whole -m gpu suite, smoke, default bench + configs[1] bench
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 120 python tests/gpu_option_fuzz.py 10 1 > $O/s11_fuzz10.txt 2>&1; echo "fuzz10 rc $?"
tail -n 3 $O/s11_fuzz10.txt | cut -c1-400
PROBE_QUICK=1 timeout 300 python tests/gpu_engine_probe.py 2000000 1536 ip 32 128 256 > $O/s11_probe1536_r2.txt 2>&1
VSS_LIBRARY=$R/duckdb-vss_amd/libvssgpu_r6.so PROBE_QUICK=1 timeout 300 python tests/gpu_engine_probe.py 2000000 1536 ip 32 128 256 > $O/s11_probe1536_r4.txt 2>&1
grep -h 'one probe\|single query' $O/s11_probe1536_r2.txt $O/s11_probe1536_r4.txt | cut -c1-330
rm -f $O/config_tests.txt
timeout 900 python -m pytest tests -q -m gpu -x --durations=8 > $O/s11_tests.txt 2>&1; echo "pytest rc $?"
tail -n 14 $O/s11_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/s11_smoke.txt 2>&1; tail -n 2 $O/s11_smoke.txt
timeout 600 python bench.py > $O/s11_bench.json 2> $O/s11_bench.err; echo "bench rc $?"
timeout 300 python bench.py --config c2 > $O/s11_bench_c2.json 2> $O/s11_bench_c2.err; echo "bench c2 rc $?"
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for f in ("s11_bench.json", "s11_bench_c2.json"):
    try:
        d = json.loads(open(O + "/" + f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("recall_at_10"), d["roofline"].get("frac"), d["roofline"].get("avg_kernel_ms"),
              (d["roofline"].get("other_regime") or {}).get("avg_kernel_ms"), d.get("host_api_queries_per_s"), d.get("build_rows_per_s"))
    except Exception as e:
        print(f, "unreadable", e)
PY
