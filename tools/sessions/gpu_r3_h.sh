#!/bin/bash
# round 3, GPU session H: build — occupancy pinned to 4 waves per SIMD, batch cap 16384 vs 32768
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config and not full_benchmark") > $O/r3h_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r3h_pytest.txt
timeout 200 python tools/gpu_solo_probe.py 1000000 128 l2sq 16 128 64 > $O/r3h_solo_1m128.txt 2>&1; grep -E "single|1024|build" $O/r3h_solo_1m128.txt
PROBE_RECALL=1 timeout 300 python tools/gpu_build_probe.py 10000000 > $O/r3h_build_default.json 2> $O/r3h_build_default.err; cat $O/r3h_build_default.json
PROBE_RECALL=1 timeout 300 python tools/gpu_build_probe.py 10000000 32768 > $O/r3h_build_32k.json 2> $O/r3h_build_32k.err; cat $O/r3h_build_32k.json
