#!/bin/bash
# One round of rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/sessions/gpu_profile_round.sh <tag> [plain|trace-only]   -> gpurun_out/prof_<tag>/summary/*   (copy those into profiles/)
# The rocpd databases are summarised on the box by tools/rocprof_summarize.py and then deleted: they are too big to
# travel back.  "plain" also runs the un-profiled default bench (with the CPU baseline) first.
set -x
ulimit -c 0
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$2" = "plain" ]; then
  python $R/bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
fi
rocprofv3 --kernel-trace --stats -d $OUT/kt -o bench -- python $R/bench.py --no-cpu-baseline --host-api-seconds 0 > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
if [ "$2" = "trace-only" ]; then
  cd $R && python tools/rocprof_summarize.py $OUT $TAG $OUT/summary > $OUT/summary.txt 2>&1
  tail -5 $OUT/summary.txt
  rm -rf $OUT/kt
  exit 0
fi
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --kernel-include-regex k_search -d $OUT/pmc_$c -o pmc -- python $R/bench.py --steps 8 --pipeline 1 --coalesce 1 --ef 96 --no-cpu-baseline --host-api-seconds 0 > $OUT/bench_pmc_$c.json 2> $OUT/pmc_$c.err
done
# MFMA kernel of the exact path: duration (kernel trace) and matrix-core busy cycles (separate PMC pass)
rocprofv3 --kernel-trace --stats -d $OUT/exact_kt -o exact -- python $R/tools/gpu_exact_probe.py 1000000 > $OUT/exact_plain.txt 2> $OUT/exact_kt.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex k_exact_scores -d $OUT/exact_pmc -o pmc -- python $R/tools/gpu_exact_probe.py 1000000 > $OUT/exact_pmc.txt 2> $OUT/exact_pmc.err
cd $R && python tools/rocprof_summarize.py $OUT $TAG $OUT/summary > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/exact_kt $OUT/exact_pmc
du -sh $R/gpurun_out
