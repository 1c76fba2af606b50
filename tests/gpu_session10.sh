#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/exact_kt -o exact -- python $R/tests/gpu_exact_probe.py 1000000 > $O/exact_plain.txt 2> $O/exact_kt.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex k_exact_scores -d $O/exact_pmc -o pmc -- python $R/tests/gpu_exact_probe.py 1000000 > $O/exact_pmc.txt 2> $O/exact_pmc.err
cd $R && python - <<'PY'
import sqlite3, json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r02b"
d = sqlite3.connect(O + "/exact_kt/exact_results.db")
full = d.execute("select avg(end-start), count(*) from kernels where name like '%k_exact_scores%' and (end-start) > 0.95 * (select max(end-start) from kernels where name like '%k_exact_scores%')").fetchone()
flops = 2.0 * 1024 * 32768 * 768
out = {"k_exact_scores_full_chunk_avg_ns": full[0], "full_chunk_launches": full[1], "tflops": flops / full[0] / 1e3, "frac_of_peak": flops / full[0] / 1e3 / 157.3}
dp = sqlite3.connect(O + "/exact_pmc/pmc_results.db")
for name, n, mean in dp.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%k_exact_scores%' group by counter_name"):
    out["pmc_%s_mean" % name] = mean
rows = d.execute("select name, count(*), avg(end-start) from kernels group by name order by 3 desc").fetchall()
out["kernels"] = [(r[0][:60], r[1], r[2]) for r in rows[:6]]
json.dump(out, open(O + "/exact_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
cat $O/exact_plain.txt | tail -2
rm -rf $O/exact_kt $O/exact_pmc
