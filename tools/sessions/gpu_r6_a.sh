#!/bin/bash
# round 6, session A: what the round's first half added, on the GPU — the co-ranking merge kernel (tests + timing at 8 x 100 x
# 32768), vss_set_option through the tests that force the engine's shapes, the build-quality test, a13 at DuckDB's call shape,
# the quality study at 200k x 768, the reference-default index with the wide ef sweep, the configs[4]-shaped union.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -m gpu -x -q -p no:cacheprovider \
   -k "merge_topk or variants_agree or compact_visited or both_engine_shapes or array_function or pipelined_contexts or sharded") > $O/r6a_pytest.txt 2>&1
echo "pytest rc $?"; tail -n 5 $O/r6a_pytest.txt
(time timeout 300 python tools/gpu_merge_probe.py) > $O/r6a_merge_probe.txt 2>&1; tail -n 8 $O/r6a_merge_probe.txt
(time timeout 300 python bench.py --config a13 --sidecar $O/r6a_a13_sidecar.json) > $O/r6a_a13.jsonl 2> $O/r6a_a13.err; tail -n 2 $O/r6a_a13.jsonl | cut -c1-1500
(time timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -p no:cacheprovider -s -k "build_quality") > $O/r6a_quality_test.txt 2>&1
echo "quality test rc $?"; tail -n 12 $O/r6a_quality_test.txt | cut -c1-900
(time timeout 900 python bench.py --config quality --sidecar $O/r6a_quality_sidecar.json) > $O/r6a_quality.jsonl 2> $O/r6a_quality.err; tail -n 1 $O/r6a_quality.jsonl | cut -c1-3000; tail -n 3 $O/r6a_quality.err | cut -c1-600
(time timeout 600 python bench.py --config c3 --M 16 --ef-construction 128 --extras none --steps 20 --warmup 5 --no-cpu-baseline --regimes none \
   --host-api-seconds 0 --no-small-launches --heldout-batches 4 --wide-ef-sweep --repeats 0 --sidecar $O/r6a_refdefault_sidecar.json) > $O/r6a_refdefault.jsonl 2> $O/r6a_refdefault.err
tail -n 1 $O/r6a_refdefault.jsonl | cut -c1-2500; grep ef_sweep $O/r6a_refdefault.jsonl | cut -c1-1200
(time timeout 1200 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -p no:cacheprovider -s -k "config4_union") > $O/r6a_config4_union.txt 2>&1
echo "config4 union rc $?"; tail -n 8 $O/r6a_config4_union.txt | cut -c1-1200
