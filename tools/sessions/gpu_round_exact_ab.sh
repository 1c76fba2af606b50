#!/bin/bash
# A/B builds of the exact path's MFMA score kernel (libvssgpu_<variant>.so next to the default library):
#   bash tools/sessions/gpu_round_exact_ab.sh default x4 bk16 ...
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02d
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  LIB=$R/duckdb-vss_amd/libvssgpu.so
  [ $v != default ] && LIB=$R/duckdb-vss_amd/libvssgpu_$v.so
  VSS_LIBRARY=$LIB timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$v -o exact -- python $R/tools/gpu_exact_probe.py 1000000 > $O/exact_$v.txt 2> $O/exact_$v.err
  echo "$v: $(tail -n 1 $O/exact_$v.txt)"
  VSS_LIBRARY=$LIB timeout 300 python -m pytest $R/tests/test_gpu_parity.py -q -m gpu -x -k "exact" -p no:cacheprovider 2>&1 | tail -n 1
done
cd $R && python - "$@" <<'PY'
import sqlite3, json, os, sys
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r02d"
out = {}
flops = 2.0 * 1024 * 32768 * 768
for v in sys.argv[1:]:
    d = sqlite3.connect(O + "/kt_%s/exact_results.db" % v)
    full = d.execute("select avg(end-start), count(*) from kernels where name like '%k_exact_scores%' and (end-start) > 0.95 * (select max(end-start) from kernels where name like '%k_exact_scores%')").fetchone()
    out[v] = {"k_exact_scores_full_chunk_avg_ns": full[0], "full_chunk_launches": full[1], "tflops": flops / full[0] / 1e3,
              "frac_of_peak": flops / full[0] / 1e3 / 157.3}
    print(v, "%.1f TFLOP/s = %.3f of peak" % (out[v]["tflops"], out[v]["frac_of_peak"]))
json.dump(out, open(O + "/exact_ab_%s.json" % "_".join(sys.argv[1:]), "w"), indent=1)
PY
rm -rf $O/kt_*
