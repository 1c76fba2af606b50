#!/bin/bash
# A/B builds of the search engine (libvssgpu_<variant>.so): single-query latency at configs[1] shape and batch timings
#   bash tools/sessions/gpu_round_engine_ab.sh default sl00 ...
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02e
mkdir -p $O
cd $R
for v in "$@"; do
  LIB=$R/duckdb-vss_amd/libvssgpu.so
  [ $v != default ] && LIB=$R/duckdb-vss_amd/libvssgpu_$v.so
  VSS_LIBRARY=$LIB PROBE_QUICK=1 timeout 200 python tools/gpu_engine_probe.py 1000000 128 l2sq 16 128 64 > $O/c2_$v.txt 2>&1
  VSS_LIBRARY=$LIB PROBE_QUICK=1 timeout 200 python tools/gpu_engine_probe.py 2000000 768 cosine 32 256 96 > $O/c3_$v.txt 2>&1
  echo "== $v"; grep -h 'one probe\|single query' $O/c2_$v.txt $O/c3_$v.txt | cut -c1-250
done
if [ -n "$AB_PARITY" ]; then  # parity subset on the last variant
  VSS_LIBRARY=$LIB timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -q -m gpu -x -p no:cacheprovider -k "not full_benchmark and not two_rank" 2>&1 | tail -n 2
fi
