#!/bin/bash
# round 5, session N: (1) matrix-pipe busy counters of the exact tile, v2 (rounds 3-4) against v4 (round 5): rocprofv3 --kernel-trace
# --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; (2) the bench's default 128 steps with the launch regimes.
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
P=$O/prof_r05_exact
mkdir -p $P
cd /tmp
export TMPDIR=/tmp
for v in 2 5; do
  VSS_EXACT_KERNEL=$v timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex k_exact_scores -d $P/pmc_$v -o pmc -- python $R/tools/gpu_exact_probe.py 2000000 > $P/exact_pmc_$v.txt 2> $P/exact_pmc_$v.err; echo "pmc kernel $v rc $?"
done
cd $R && python - <<'PY'
import glob, json, os, sqlite3
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
P = R + "/gpurun_out/prof_r05_exact"
out = {"workload": "1024 queries x 2000000 rows x FLOAT[768] cosine (tools/gpu_exact_probe.py): the select folded into the tile, windows of 8 x 32768 rows",
       "command": "VSS_EXACT_KERNEL={2|5} rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex k_exact_scores -- python tools/gpu_exact_probe.py 2000000",
       "mfma_busy_frac": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 * 4): GRBM_GUI_ACTIVE is summed over the 8 XCDs, the MFMA counter over all SIMDs",
       "peak_tflops_f32_matrix": 157.3}
for v in ("2", "5"):
    d = sqlite3.connect(sorted(glob.glob(P + "/pmc_%s/**/*.db" % v, recursive=True))[0])
    e = {"wall": [l.strip() for l in open(P + "/exact_pmc_%s.txt" % v) if "exact top" in l][-1:]}
    name = d.execute("select distinct kernel_name from counters_collection where kernel_name like '%k_exact_scores%'").fetchall()
    e["kernel"] = name[0][0][:60] if name else None
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
        rows = d.execute("select sum(value) from counters_collection where counter_name = ? and kernel_name like '%k_exact_scores%' group by dispatch_id", (c,)).fetchall()
        e["pmc_%s_sum_over_launches" % c] = sum(r[0] for r in rows)
        e["pmc_launches"] = len(rows)
    e["mfma_busy_frac"] = e["pmc_SQ_VALU_MFMA_BUSY_CYCLES_sum_over_launches"] / (e["pmc_GRBM_GUI_ACTIVE_sum_over_launches"] / 8 * 256 * 4)
    ks = d.execute("select sum(end-start), count(*) from kernels where name like '%k_exact_scores%'").fetchone()
    e["kernel_total_ms_under_counters"] = ks[0] / 1e6
    e["tflops_under_counters"] = 4 * 2.0 * 1024 * 2000000 * 768 / (ks[0] * 1e-9) / 1e12
    out["kernel_%s" % v] = e
json.dump(out, open(R + "/gpurun_out/r05_exact_mfma_busy.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $P/pmc_2 $P/pmc_5
timeout 900 python bench.py --no-cpu-baseline --host-api-seconds 0 --extras none --no-small-launches --regimes 4x1,8x1,16x1,8x2,8x3 --sidecar $O/r05_bench_default_128_steps_regimes_sidecar.json > $O/r05_bench_default_128_steps_regimes.jsonl 2> $O/r05_default.err; echo "default bench rc $?"
grep '"regime"\|"metric"' $O/r05_bench_default_128_steps_regimes.jsonl | cut -c1-330
