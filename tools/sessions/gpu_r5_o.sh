#!/bin/bash
# round 5, session O: up to 32 batches per launch (MAX_COALESCED 16 -> 32) — the collected tests of the multi-batch paths, then the
# round's evidence run once more (gpu_r5_final.sh nosuite: the driver's command plain / under rocprofv3 / two PMC passes).
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -m gpu -x -q -p no:cacheprovider \
   -k "several_batches or pipelined_contexts or concurrent or two_rank or one_rank or variants_agree") > $O/r5o_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r5o_pytest.txt
bash tools/sessions/gpu_r5_final.sh nosuite
