#!/bin/bash
# round 5, session P: the whole -m gpu suite and smoke() on the round's FINAL tree (after MAX_COALESCED went to 32).
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 -p no:cacheprovider) > $O/r5_final_tests.txt 2>&1; echo "pytest rc $?"
tail -n 14 $O/r5_final_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r5_final_smoke.txt 2>&1; tail -n 2 $O/r5_final_smoke.txt
