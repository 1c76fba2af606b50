#!/bin/bash
# round 5, session M: queries that outgrow their LDS-resident visited set repeated in place (SearchArgs::retry_hash) against the
# host's second launch; wall clock per call (re-runs included), 768 and 1536 dims; then the collected tests of the touched paths.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
timeout 600 python tools/gpu_wide_list_probe.py 10000000 768 cosine 16 128 10 512,288 > $O/r5m_retry_in_place_10m768.txt 2>&1; echo "probe 768 rc $?"; grep -v "^built\|amdgpu.ids" $O/r5m_retry_in_place_10m768.txt
timeout 600 python tools/gpu_wide_list_probe.py 3000000 1536 ip 32 128 100 480 > $O/r5m_retry_in_place_3m1536.txt 2>&1; echo "probe 1536 rc $?"; grep -v "^built\|amdgpu.ids" $O/r5m_retry_in_place_3m1536.txt
(time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -m gpu -x -q -p no:cacheprovider \
   -k "variants_agree or fuzz or compact_visited or limits_beyond or register_queue or several_batches or pipelined or filtered or concurrent") > $O/r5m_pytest.txt 2>&1
echo "pytest rc $?"; tail -n 4 $O/r5m_pytest.txt
