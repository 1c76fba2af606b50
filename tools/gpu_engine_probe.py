"""Search-engine shape sweep on bench-shaped data (not a pytest module):

    python tools/gpu_engine_probe.py [rows=10000000] [dim=768] [metric=cosine] [M=32] [efc=256] [ef=96]

Builds once, then times 1024-query batches (one probe at a time, and three in flight) for several (waves, walkers)
shapes, and the single-query entry point; prints kernel ms, queries/s and the algorithmic HBM rate per launch.
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

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
metric = sys.argv[3] if len(sys.argv) > 3 else "cosine"
M = int(sys.argv[4]) if len(sys.argv) > 4 else 32
efc = int(sys.argv[5]) if len(sys.argv) > 5 else 256
ef = int(sys.argv[6]) if len(sys.argv) > 6 else 96
pkg = load_package()
dev = torch.device("cuda", 0)
gen = bench.Mixture(rows, dim, metric != "l2sq", dev)
idx = pkg.GpuIndex(dim, metric, M, 2 * M, efc)
idx.reserve(rows)
pos = 0
while pos < rows:
    n = min(bench.CHUNK, rows - pos)
    x = gen.rows(bench.DATA_SEED, pos // bench.CHUNK, bench.CHUNK)[:n].contiguous()
    ids = torch.arange(pos, pos + n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    idx.stage_device(ids.data_ptr(), x.data_ptr(), n)
    pos += n
    del x, ids
t0 = time.time()
idx.build_finalize()
print("build %d x %d %s: %.2fs = %.0f rows/s" % (rows, dim, metric, time.time() - t0, rows / (time.time() - t0)), flush=True)
k, B = 10, 1024
Q = [gen.rows(bench.QUERY_SEED, i, B) for i in range(4)]
outs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
         torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(4)]
tk = torch.empty((B, k), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
idx.search_batch_device(Q[0].data_ptr(), B, k, 0, tk.data_ptr(), outs[0][1].data_ptr(), outs[0][2].data_ptr(), exact=True)
truth = tk.clone()
ref = None
quick = os.environ.get("PROBE_QUICK") == "1"  # default shape only
for waves, walkers, la in (((16, 0, 0),) if quick else ((16, 4, 0), (16, 4, 1), (16, 4, 2), (16, 4, 4), (12, 4, 2), (16, 3, 2), (16, 0, 2))):
    idx.set_search_params(waves, walkers)
    idx.set_search_lookahead(la)
    a, b, c = outs[0]
    ms = []
    for i in range(6):
        idx.search_batch_device(Q[i % 4].data_ptr(), B, k, ef, a.data_ptr(), b.data_ptr(), c.data_ptr())
        ms.append(idx.timing()["search_kernel_ms"])
    idx.search_batch_device(Q[0].data_ptr(), B, k, ef, a.data_ptr(), b.data_ptr(), c.data_ptr())
    st = idx.last_search_stats()
    rec = bench.recall_at_k(a, truth)
    if ref is None:
        ref = a.clone()
    same = bool(torch.equal(ref, a))
    gb = (st[0] * (4 * dim + 4) + st[1] * (4 + 8 * M)) / 1e9
    best = min(ms[1:])
    # three probes in flight
    torch.cuda.synchronize()
    steps = 24
    t0 = time.perf_counter()
    for i in range(steps + 3):
        cc = i % 3
        if i >= 3:
            idx.search_end(cc)
        if i < steps:
            o = outs[cc]
            idx.search_begin(cc, Q[i % 4].data_ptr(), B, k, ef, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    print("waves %2d walkers %d look-ahead %d: one probe %.3f ms (%s) = %.0f q/s, %.0f GB/s = %.3f of 8 TB/s; three in flight %.3f ms/step = "
          "%.0f q/s, %.0f GB/s over wall; recall %.4f same_ids %s dists/q %.0f exp/q %.0f" % (
              waves, walkers, la, best, ",".join("%.2f" % m for m in ms), B / best * 1e3, gb / (best / 1e3), gb / (best / 1e3) / 8000,
              wall * 1e3, B / wall, gb / wall, rec, same, st[0] / B, st[1] / B), flush=True)
# single-query entry point
Qh = Q[1].cpu().numpy()
for waves, walkers, la in (((16, 1, 0),) if quick else ((16, 1, 0), (16, 1, 1), (8, 1, 1))):
    idx.set_search_params(waves, walkers)
    idx.set_search_lookahead(la)
    for i in range(8):
        idx.search(Qh[i], k, ef)
    t0 = time.perf_counter()
    for i in range(500):
        idx.search(Qh[i], k, ef)
    dt = (time.perf_counter() - t0) / 500
    print("single query, waves %2d look-ahead %d: %.1f us/call = %.0f q/s (kernel %.1f us)" % (waves, la, dt * 1e6, 1 / dt,
                                                                               idx.timing()["search_kernel_ms"] * 1e3), flush=True)
