#!/bin/bash
# round 4, last GPU session: the sizing rule moved into host_logic.h (refactor) — parity subset of everything that sizes a visited set
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "variants or wide_lists or limits_beyond or both_engine_shapes or option_space_fuzz or register_queue or several_batches or tombstones or reference_built or bulk_build") > $O/r4_last_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 5 $O/r4_last_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1
