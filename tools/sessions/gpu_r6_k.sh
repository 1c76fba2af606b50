#!/bin/bash
# round 6, session K: ONE batch per launch (1024 queries, 256 compute units) with fewer walkers per workgroup than queries per
# compute unit — walkers then pull their next query from the launch's counter when they finish one, instead of every walker
# holding exactly one query from start to end (the launch lasts as long as the compute unit with the longest four).
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
for w in 0 1 2 3; do
  VSS_SEARCH_WALKERS=$w timeout 400 python tools/gpu_crew_probe.py 3000000 768 cosine 32 256 80 2>&1 | grep -v amdgpu > $O/r6k_crew_probe_3m768_walkers_$w.txt; echo "walkers=$w rc $?"
  grep "^B=1024\|^B= 256\|x 1024 queries per launch, crews+pipe plain" $O/r6k_crew_probe_3m768_walkers_$w.txt | cut -c1-260
done
