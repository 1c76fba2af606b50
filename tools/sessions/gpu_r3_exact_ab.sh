#!/bin/bash
# round 3: the MFMA score tile, round 2's kernel (VSS_EXACT_KERNEL=1) against the software-pipelined one (=2, default)
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03_exact
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 python -m pytest $R/tests -q -m gpu -x -k "exact or array_ or readme" -p no:cacheprovider 2>&1 | tail -n 2
for v in 1 2; do
  VSS_EXACT_KERNEL=$v timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$v -o exact -- python $R/tools/gpu_exact_probe.py 1000000 > $O/exact_$v.txt 2> $O/exact_$v.err
  echo "kernel $v: $(tail -n 1 $O/exact_$v.txt)"
  VSS_EXACT_KERNEL=$v timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex k_exact_scores -d $O/pmc_$v -o pmc -- python $R/tools/gpu_exact_probe.py 1000000 > $O/exact_pmc_$v.txt 2> $O/exact_pmc_$v.err
done
cd $R && python - <<'PY'
import sqlite3, json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r03_exact"
out = {"workload": "1024 queries x 1000000 rows x FLOAT[768] cosine, 32768-row chunks: one score tile launch = 1024 x 32768 x 768",
       "command": "VSS_EXACT_KERNEL={1|2} rocprofv3 --kernel-trace {--stats | --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE "
                  "--kernel-include-regex k_exact_scores} -- python tools/gpu_exact_probe.py 1000000", "peak_tflops_f32_matrix": 157.3}
flops = 2.0 * 1024 * 32768 * 768
for v in ("1", "2"):
    try:
        d = sqlite3.connect(O + "/kt_%s/exact_results.db" % v)
        full = d.execute("select avg(end-start), count(*), min(name) from kernels where name like '%k_exact_scores%' and (end-start) > 0.95 * (select max(end-start) from kernels where name like '%k_exact_scores%')").fetchone()
        e = {"kernel": full[2][:60], "full_chunk_avg_ns": full[0], "full_chunk_launches": full[1], "tflops": flops / full[0] / 1e3, "frac_of_peak": flops / full[0] / 1e3 / 157.3}
        e["wall"] = open(O + "/exact_%s.txt" % v).read().strip().splitlines()[-1]
        dp = sqlite3.connect(O + "/pmc_%s/pmc_results.db" % v)
        for name, n, mean in dp.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%k_exact_scores%' group by counter_name"):
            e["pmc_%s_mean" % name], e["pmc_launches"] = mean, n
        if "pmc_SQ_VALU_MFMA_BUSY_CYCLES_mean" in e and "pmc_GRBM_GUI_ACTIVE_mean" in e:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles per XCD = value / 8; the MFMA counter is summed over all SIMDs
            e["mfma_busy_frac"] = e["pmc_SQ_VALU_MFMA_BUSY_CYCLES_mean"] / (e["pmc_GRBM_GUI_ACTIVE_mean"] / 8 * 256 * 4)
        out["kernel_%s" % v] = e
        print(v, json.dumps(e))
    except Exception as ex:
        print(v, "failed:", ex)
json.dump(out, open(O + "/exact_ab.json", "w"), indent=1)
PY
rm -rf $O/kt_* $O/pmc_*
