"""Round 3, locality: the same index before and after vss_compact's (level, cluster) reordering.
   python tools/gpu_locality_probe.py [rows] [dim] [metric] [M] [efc] [ef]
Prints queries/s, k_search ms per launch and the algorithmic GB/s per launch for 16-batch launches (one at a time and
three gated in flight), recall@10 against the exact path, and what the compaction cost."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
metric = sys.argv[3] if len(sys.argv) > 3 else "cosine"
M = int(sys.argv[4]) if len(sys.argv) > 4 else 32
efc = int(sys.argv[5]) if len(sys.argv) > 5 else 256
ef = int(sys.argv[6]) if len(sys.argv) > 6 else 96
k, B, G = 10, 1024, 16
dev = torch.device("cuda", 0)
pkg = load_package()
gen = bench.Mixture(rows, dim, metric != "l2sq", dev)
idx = pkg.GpuIndex(dim, metric, M, 2 * M, efc, ef)
idx.reserve(rows)
for c in range(0, rows, bench.CHUNK):
    m = min(bench.CHUNK, rows - c)
    x = gen.rows(bench.DATA_SEED, c // bench.CHUNK, m)
    ids = torch.arange(c, c + m, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    idx.stage_device(ids.data_ptr(), x.data_ptr(), m)
    del x, ids
t0 = time.perf_counter()
idx.build_finalize()
print("build %.2f s (%.0f rows/s)" % (time.perf_counter() - t0, rows / (time.perf_counter() - t0)), flush=True)
Q = [gen.rows(bench.QUERY_SEED, i, B) for i in range(3 * G)]
outs = [[(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
          torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(G)] for _ in range(3)]
truth = torch.empty((B, k), dtype=torch.int64, device=dev)
idx.search_batch_device(Q[0].data_ptr(), B, k, 0, truth.data_ptr(), outs[0][0][1].data_ptr(), outs[0][0][2].data_ptr(), exact=True)
torch.cuda.synchronize()


def begin(c, j):
    idx.search_multi_begin(c, [Q[(j * G + i) % len(Q)].data_ptr() for i in range(G)], B, k, ef, [o[0].data_ptr() for o in outs[c]],
                           [o[1].data_ptr() for o in outs[c]], [o[2].data_ptr() for o in outs[c]])


def measure(tag, depth, launches=6):
    for c in range(depth):
        begin(c, c)
    for c in range(depth):
        idx.search_end(c)
    torch.cuda.synchronize()
    kms, nd, ne = 0.0, 0, 0
    t0 = time.perf_counter()
    for j in range(launches + depth):
        c = j % depth
        if j >= depth:
            idx.search_end(c)
            kms += idx.timing()["search_kernel_ms"]
            st = idx.last_search_stats()
            nd, ne = nd + int(st[0]), ne + int(st[1])
        if j < launches:
            begin(c, j)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    by = nd * (4 * dim + 4) + ne * (4 + 8 * M)
    print("%-34s %dx%d: %8.0f q/s, %.2f ms per launch, %.0f GB/s per launch (%.3f of 8 TB/s), %.0f GB/s over wall; %.1f dist %.1f exp per query"
          % (tag, G, depth, launches * G * B / dt, kms / launches, by / (kms / 1e3) / 1e9, by / (kms / 1e3) / 1e9 / 8000, by / dt / 1e9,
             nd / (launches * G * B), ne / (launches * G * B)), flush=True)


def recall():
    idx.search_batch_device(Q[0].data_ptr(), B, k, ef, outs[0][0][0].data_ptr(), outs[0][0][1].data_ptr(), outs[0][0][2].data_ptr())
    torch.cuda.synchronize()
    return bench.recall_at_k(outs[0][0][0], truth), outs[0][0][0].cpu().numpy().copy(), outs[0][0][1].cpu().numpy().copy()


r0, k0, d0 = recall()
measure("insertion order", 1)
measure("insertion order", 3)
t0 = time.perf_counter()
done = idx.compact(True)
t_c = time.perf_counter() - t0
r1, k1, d1 = recall()
print("vss_compact (reordered=%s): %.2f s; recall@10 %.4f -> %.4f; answers identical: ids %s, distance bits %s"
      % (done, t_c, r0, r1, np.array_equal(k0, k1), np.array_equal(d0.view(np.uint32), d1.view(np.uint32))), flush=True)
measure("(level, cluster) order", 1)
measure("(level, cluster) order", 3)
