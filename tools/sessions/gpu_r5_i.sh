#!/bin/bash
# round 5, session I: the whole -m gpu suite and smoke() on the tree with the round's defaults (12-wave pipelined wide lists,
# k_exact_scores_v4, exact filter windows sized by k).
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider) > $O/r5i_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 6 $O/r5i_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r5i_smoke.txt 2>&1; echo "smoke rc $?"; tail -n 2 $O/r5i_smoke.txt
