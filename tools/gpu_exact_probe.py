"""Exact (brute-force, MFMA f32) search workload for rocprofv3 runs (not a pytest module):
1024 queries x [rows] x 768 through vss_search_exact_batch_device; prints TFLOP/s from the engine's own timers."""
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

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim, metric, B, k = 768, "cosine", 1024, 10
pkg = load_package()
dev = torch.device("cuda", 0)
gen = bench.Mixture(rows, dim, True, dev)
idx = pkg.GpuIndex(dim, metric, 4, 8, 8)  # the graph is irrelevant here: cheapest possible build
idx.reserve(rows)
x = gen.rows(bench.DATA_SEED, 0, rows)
ids = torch.arange(rows, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
idx.stage_device(ids.data_ptr(), x.data_ptr(), rows)
idx.build_finalize()
q = gen.rows(bench.QUERY_SEED, 0, B)
ok = torch.empty((B, k), dtype=torch.int64, device=dev)
od = torch.empty((B, k), dtype=torch.float32, device=dev)
oc = torch.empty(B, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
idx.search_batch_device(q.data_ptr(), B, k, 0, ok.data_ptr(), od.data_ptr(), oc.data_ptr(), exact=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    idx.search_batch_device(q.data_ptr(), B, k, 0, ok.data_ptr(), od.data_ptr(), oc.data_ptr(), exact=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("exact top-%d, %d queries x %d rows x %d: %.4f s per batch = %.1f TFLOP/s over wall (scores + select + re-rank)" % (
    k, B, rows, dim, dt, 2.0 * B * rows * dim / dt / 1e12))
