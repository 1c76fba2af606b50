#!/bin/bash
# whole suite, smoke, the driver's bench command, the default bench, rocprof trace of the bench, configs[1] line
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
P=$O/prof_r02c
mkdir -p $P/summary
cd $R
rm -f $O/config_tests.txt
timeout 900 python -m pytest tests -q -m gpu -x --durations=5 > $O/s14_tests.txt 2>&1; echo "pytest rc $?"
tail -n 9 $O/s14_tests.txt
cat $O/config_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/s14_smoke.txt 2>&1; tail -n 2 $O/s14_smoke.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $P/bench_plain.json 2> $P/bench_plain.err; echo "driver-style bench rc $?"
if [ "$1" = "quick" ]; then
  timeout 300 python bench.py --config c2 > $O/s14_bench_c2.json 2> $O/s14_bench_c2.err; echo "bench c2 rc $?"
  python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
d = json.loads([l for l in open(O + "/prof_r02c/bench_plain.json").read().splitlines() if l.startswith("{")][-1])
r = d["roofline"]
print("driver-style: value %.0f ms/step %.3f frac/launch %.3f kernel ms %.3f over wall %.3f recall %s host_api %s build %.0f exact %.3f" % (
    d["value"], d["ms_per_step"], r["frac"], r["avg_kernel_ms"], r["frac_over_wall"], d["recall_at_10"], d.get("host_api_queries_per_s"),
    d["build_rows_per_s"], d["exact_batch_s"]))
d = json.loads(open(O + "/s14_bench_c2.json").read().strip().splitlines()[-1])
print("c2:", d["value"], d["ms_per_step"], d["roofline"].get("us_per_expansion"), d["cpu_baseline"]["value"])
PY
  exit 0
fi
timeout 600 python bench.py --no-cpu-baseline --host-api-seconds 0 --regimes 8x1,8x2,4x3,8x3u > $O/s14_bench_default.json 2> $O/s14_bench_default.err; echo "default bench rc $?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $P/kt -o bench -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --ef 96 --regimes none --no-cpu-baseline --host-api-seconds 0 > $P/bench_under_rocprof.json 2> $P/kt.err; echo "rocprof rc $?"
cd $R && python tools/rocprof_summarize.py $P r02c $P/summary > $P/summary.txt 2>&1
tail -n 4 $P/summary.txt | cut -c1-600
rm -rf $P/kt
timeout 300 python bench.py --config c2 > $O/s14_bench_c2.json 2> $O/s14_bench_c2.err; echo "bench c2 rc $?"
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for f in ("prof_r02c/bench_plain.json", "s14_bench_default.json", "prof_r02c/bench_under_rocprof.json"):
    d = json.loads([l for l in open(O + "/" + f).read().splitlines() if l.startswith("{")][-1])
    r = d["roofline"]
    print("%s: %dx%d steps %d value %.0f ms/step %.3f frac/launch %.3f kernel ms %.3f (%d launches) over wall %.3f recall %s host_api %s build %.0f" % (
        f, d["config"]["batches_per_launch"], d["config"]["launches_in_flight"], d["steps"], d["value"], d["ms_per_step"], r["frac"],
        r["avg_kernel_ms"], r["launches"], r["frac_over_wall"], d["recall_at_10"], d.get("host_api_queries_per_s"), d["build_rows_per_s"]))
    for g in r["regimes"]:
        print("  %dx%d%s: %.0f q/s, %.3f ms/step, launch %.3f ms, frac/launch %.3f, over wall %.3f" % (
            g["batches_per_launch"], g["launches_in_flight"], "" if g["gated"] else "u", g["queries_per_s"], g["ms_per_step"],
            g["avg_kernel_ms"], g["frac_per_launch"], g["frac_over_wall"]))
d = json.loads(open(O + "/s14_bench_c2.json").read().strip().splitlines()[-1])
print("c2:", d["value"], d["ms_per_step"], d["roofline"].get("us_per_expansion"), d["cpu_baseline"]["value"])
PY
