#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r3f_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3f_pytest.txt
VSS_SEARCH_SOLO=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark and not two_rank and not variants" > $O/r3f_pytest_solo_forced.txt 2>&1; echo "solo-forced pytest rc $?"; tail -n 3 $O/r3f_pytest_solo_forced.txt
