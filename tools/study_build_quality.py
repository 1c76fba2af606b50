"""Build-quality study on the CPU (not a pytest module): does the engine's batch-synchronous schedule
build a graph as good as the reference's own build at configs[1] size?

    python tools/study_build_quality.py [rows=1000000] [dim=128] [out=profiles/r02_build_quality.json]

For each data spec (bench.py's mixture with centre scale 0.1, and SURVEY §8d's original centre scale 1.0):
  A  the REFERENCE library (oracle/_ref, the vendored usearch) building with its own threading model, several add()
     streams (hnsw_index_physical_create.cpp:239-245);
  B  the restatement in KERNEL mode (wave summation order, kernel candidate lists) with the engine's batch schedule
     batch = clamp(nodes / 32, 1, 16384) — byte for byte the graph the GPU builds (tests/test_gpu_parity.py);
then recall@10 of both against exact brute force at ef_search 64 / 128 with the reference defaults (M=16, M0=32,
ef_construction=128), same queries.  tests/test_host_logic.py asserts batched >= reference - 0.015 on the committed table.
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import bench  # noqa: E402
from oracle_lib import CpuIndex, load_oracle, load_ref  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r02_build_quality.json")
threads = int(os.environ.get("STUDY_THREADS", "4"))
growth_div = int(os.environ.get("STUDY_GROWTH_DIV", "32"))  # batch = clamp(nodes / growth_div, 1, 16384)
scales = [float(x) for x in os.environ.get("STUDY_SCALES", "0.1,1.0").split(",")]
metric, M, M0, efc, k, nq = "l2sq", 16, 32, 128, 10, 1000
ref, orc = load_ref(), load_oracle()
assert ref is not None, "needs oracle/_ref (the reference's usearch build)"
results = []
for centre_scale in scales:
    bench.CENTRE_SCALE = centre_scale
    gen = bench.Mixture(rows, dim, False, torch.device("cpu"))
    X = torch.cat([gen.rows(bench.DATA_SEED, c, min(bench.CHUNK, rows - c * bench.CHUNK))
                   for c in range((rows + bench.CHUNK - 1) // bench.CHUNK)]).numpy()
    Q = gen.rows(bench.QUERY_SEED, 0, nq).numpy()
    # exact ground truth, blockwise
    best_d = np.full((nq, k), np.inf, dtype=np.float32)
    best_i = np.zeros((nq, k), dtype=np.int64)
    q2 = (Q ** 2).sum(1)[:, None]
    for c in range(0, rows, 100_000):
        xb = X[c:c + 100_000]
        d = q2 - 2 * Q @ xb.T + (xb ** 2).sum(1)[None, :]
        cand_d = np.concatenate([best_d, d], 1)
        cand_i = np.concatenate([best_i, np.arange(c, c + len(xb))[None, :].repeat(nq, 0)], 1)
        sel = np.argpartition(cand_d, k, axis=1)[:, :k]
        best_d, best_i = np.take_along_axis(cand_d, sel, 1), np.take_along_axis(cand_i, sel, 1)

    def recall(keys):
        return float(np.mean([len(set(keys[i].tolist()) & set(best_i[i].tolist())) / k for i in range(nq)]))

    row = {"rows": rows, "dim": dim, "metric": metric, "M": M, "M0": M0, "ef_construction": efc, "centre_scale": centre_scale,
           "queries": nq}
    a = CpuIndex(ref, dim, metric, M, M0, efc, 64)
    t0 = time.time()
    a.add_mt(np.arange(rows), X, threads)
    row["reference_build_s"], row["reference_threads"] = time.time() - t0, threads
    b = CpuIndex(orc, dim, metric, M, M0, efc, 64, order=1, wave=1)
    b.reserve(rows, 1)
    t0 = time.time()
    b.build_batch(np.arange(rows), X, 16384, growth_div)
    row["growth_div"] = growth_div
    row["batched_build_s"] = time.time() - t0
    for ef in (64, 128):
        row["reference_recall_ef%d" % ef] = recall(a.search_many(Q, k, ef=ef)[0])
        row["batched_recall_ef%d" % ef] = recall(b.search_many(Q, k, ef=ef)[0])
    print(json.dumps(row), flush=True)
    results.append(row)
    del a, b, X
with open(out, "w") as f:
    json.dump({"what": "recall@10 vs exact brute force: reference usearch build (several add() streams) vs the engine's "
                       "batch-synchronous schedule (CPU restatement in kernel mode = the GPU's graph), same data, options and queries",
               "command": "python tools/study_build_quality.py %d %d" % (rows, dim), "results": results}, f, indent=1)
