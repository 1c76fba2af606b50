"""Bulk build workload for rocprofv3 --pmc runs (not a pytest module): rows x 768 cosine, the bench's index options.
Prints one JSON line with the engine's own work counters (algorithmic bytes of phase A)."""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
max_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # 0 = the engine's default schedule
efc = int(sys.argv[3]) if len(sys.argv) > 3 else 256  # (round 6: the headline's ef_construction is 384)
dim, metric, M = 768, "cosine", 32
pkg = load_package()
dev = torch.device("cuda", 0)
gen = bench.Mixture(rows, dim, True, dev)
idx = pkg.GpuIndex(dim, metric, M, 2 * M, efc)
idx.reserve(rows)
if max_batch:
    idx.set_build_params(max_batch, 32)
for c in range(0, rows, bench.CHUNK):
    m = min(bench.CHUNK, rows - c)
    x = gen.rows(bench.DATA_SEED, c // bench.CHUNK, m)
    ids = torch.arange(c, c + m, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    idx.stage_device(ids.data_ptr(), x.data_ptr(), m)
t0 = time.perf_counter()
idx.build_finalize()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
tm, work = idx.timing(), idx.build_work()
a_bytes = work["insert_distances"] * (4 * dim + 4) + work["insert_expansions"] * (4 + 4 * 2 * M)
recall = None
if os.environ.get("PROBE_RECALL"):
    import numpy as np
    B, k, ef = 1024, 10, 96
    q = gen.rows(bench.QUERY_SEED, 0, B)
    ok, tk = (torch.empty((B, k), dtype=torch.int64, device=dev) for _ in range(2))
    od = torch.empty((B, k), dtype=torch.float32, device=dev)
    oc = torch.empty(B, dtype=torch.int32, device=dev)
    idx.search_batch_device(q.data_ptr(), B, k, 0, tk.data_ptr(), od.data_ptr(), oc.data_ptr(), exact=True)
    idx.search_batch_device(q.data_ptr(), B, k, ef, ok.data_ptr(), od.data_ptr(), oc.data_ptr())
    torch.cuda.synchronize()
    recall = bench.recall_at_k(ok, tk)
print(json.dumps({"rows": rows, "max_batch": max_batch or "default", "recall_at_10_ef96": recall, "rows_per_s": rows / dt,
                  "batches": tm["build_batches"], "dim": dim, "metric": metric, "M": M, "ef_construction": efc, "build_s": dt,
                  "phase_a_ms": tm["build_phase_a_ms"], "phase_b_ms": tm["build_phase_b_ms"],
                  "phase_a_distances": work["insert_distances"], "phase_a_expansions": work["insert_expansions"],
                  "phase_b_distances": work["link_distances"],
                  "phase_a_algorithmic_bytes": a_bytes,
                  "phase_a_algorithmic_gbs": a_bytes / (tm["build_phase_a_ms"] / 1e3) / 1e9}))
