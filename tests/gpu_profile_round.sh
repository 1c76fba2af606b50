set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r1c
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $R/gpurun_out/prof_r1c/bench_plain.json 2> $R/gpurun_out/prof_r1c/bench_plain.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1c/kt -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_r1c/bench_under_rocprof.json 2> $R/gpurun_out/prof_r1c/kt.err
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c --kernel-include-regex k_search -d $R/gpurun_out/prof_r1c/pmc_$c -o pmc -- python $R/bench.py --steps 8 --pipeline 1 --ef 96 --no-cpu-baseline > $R/gpurun_out/prof_r1c/bench_pmc_$c.json 2> $R/gpurun_out/prof_r1c/pmc_$c.err
done
find $R/gpurun_out/prof_r1c -name "*.csv" | head -30
tail -1 $R/gpurun_out/prof_r1c/bench_plain.json | cut -c1-400
