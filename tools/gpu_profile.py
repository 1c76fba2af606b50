"""Kernel-time probe for the GPU box (not a pytest module): build an index on bench-shaped data, then report search
kernel milliseconds (hipEvents inside the engine) across ef and batch sizes, and the build phase split.

    python tools/gpu_profile.py [rows] [dim] [metric]
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
metric = sys.argv[3] if len(sys.argv) > 3 else "cosine"
pkg = load_package()
dev = torch.device("cuda", 0)
gen = bench.Mixture(rows, dim, metric != "l2sq", dev)
idx = pkg.GpuIndex(dim, metric)
idx.reserve(rows)
pos = 0
while pos < rows:
    n = min(bench.CHUNK, rows - pos)
    x = gen.rows(bench.DATA_SEED, pos // bench.CHUNK, bench.CHUNK)[:n].contiguous()
    ids = torch.arange(pos, pos + n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    idx.stage_device(ids.data_ptr(), x.data_ptr(), n)
    pos += n
t0 = time.time()
idx.build_finalize()
t1 = time.time()
print("build %d x %d %s: %.2fs = %.0f rows/s  %s" % (rows, dim, metric, t1 - t0, rows / (t1 - t0), idx.timing(reset=True)))
k = 10
for B in (1024, 4096, 256):
    q = gen.rows(bench.QUERY_SEED, 0, B)
    ok = torch.empty((B, k), dtype=torch.int64, device=dev)
    od = torch.empty((B, k), dtype=torch.float32, device=dev)
    oc = torch.empty(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    idx.search_batch_device(q.data_ptr(), B, k, 0, ok.data_ptr(), od.data_ptr(), oc.data_ptr(), exact=True)
    truth = ok.clone()
    for ef in (64, 128, 256):
        idx.search_batch_device(q.data_ptr(), B, k, ef, ok.data_ptr(), od.data_ptr(), oc.data_ptr())
        ms = []
        for _ in range(3):
            idx.search_batch_device(q.data_ptr(), B, k, ef, ok.data_ptr(), od.data_ptr(), oc.data_ptr())
            ms.append(idx.timing()["search_kernel_ms"])
        st = idx.last_search_stats()
        nd, ne = st[0] / B, st[1] / B
        gb = (st[0] * (4 * dim + 4) + st[1] * 132) / 1e9
        print("B=%d ef=%d: kernel %.3f ms (min of %s) -> %.0f qps, recall %.3f, dists/q %.0f, expansions/q %.0f, "
              "%.2f us/expansion/wave-round, %.0f GB/s, retried %d" % (
                  B, ef, min(ms), ["%.2f" % m for m in ms], B / min(ms) * 1e3, bench.recall_at_k(ok, truth), nd, ne,
                  min(ms) * 1e3 / ne, gb / (min(ms) / 1e3), st[3]))
