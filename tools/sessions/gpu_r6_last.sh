#!/bin/bash
# round 6, last session: the driver's command once more on the round's last tree (search contexts sized once for the largest launch:
# no reallocation inside the timed region; the reference-default extra carries its ef 512 operating point beside the wide sweep),
# and the parity files.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --ignore tests/test_gpu_configs.py) > $O/r6_last_pytest.txt 2>&1; echo "pytest rc $?"; tail -n 3 $O/r6_last_pytest.txt | cut -c1-300
(time timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 --sidecar $O/r06_bench_last_tree_sidecar.json) > $O/r06_bench_last_tree_stdout.jsonl 2> $O/r06_bench_last_tree.err; echo "driver-style bench rc $?"; tail -n 3 $O/r06_bench_last_tree.err
tail -c 8000 $O/r06_bench_last_tree_stdout.jsonl > $O/r06_bench_last_tree_last_8000_chars.txt
grep '"extra"\|"detail": "repeat"\|"detail": "regime"\|small_launches' $O/r06_bench_last_tree_stdout.jsonl | cut -c1-500
tail -n 1 $O/r06_bench_last_tree_stdout.jsonl | cut -c1-900
