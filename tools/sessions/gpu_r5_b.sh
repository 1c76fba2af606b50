#!/bin/bash
# round 5, session B: limits of 257-512 as 12-wave workgroups with the pipelined level search (vss_set_search_wide_lists) against
# round 4's 16 waves / plain order — exactness (collected tests of the touched paths) and rates at 768 and 1536 dims.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
(time timeout 900 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider \
   -k "compact_visited or variants_agree or exact_search_over_many or array_function or limits_beyond or pipelined or several_batches") > $O/r5b_pytest.txt 2>&1
echo "pytest rc $?"; tail -n 5 $O/r5b_pytest.txt
timeout 600 python tools/gpu_wide_list_probe.py 10000000 768 cosine 16 128 10 512,384,288 > $O/r5b_wide_lists_10m768.txt 2>&1; echo "probe 768 rc $?"; grep -v "^built" $O/r5b_wide_lists_10m768.txt | tail -n 14
timeout 600 python tools/gpu_wide_list_probe.py 3000000 1536 ip 32 128 100 480,320 > $O/r5b_wide_lists_3m1536.txt 2>&1; echo "probe 1536 rc $?"; grep -v "^built" $O/r5b_wide_lists_3m1536.txt | tail -n 10
