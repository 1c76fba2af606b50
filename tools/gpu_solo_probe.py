"""Engine shape A/B for small launches (round 3): the solo shape (one self-scoring wave per query) against the workgroup
shape (walkers + scoring waves) for launches of 1 .. 1024 queries.   python tools/gpu_solo_probe.py [rows] [dim] [metric] [M] [efc] [ef]"""
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

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
metric = sys.argv[3] if len(sys.argv) > 3 else "l2sq"
M = int(sys.argv[4]) if len(sys.argv) > 4 else 16
efc = int(sys.argv[5]) if len(sys.argv) > 5 else 128
ef = int(sys.argv[6]) if len(sys.argv) > 6 else 64
k = 10
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
t0 = time.perf_counter()
idx.build_finalize()
print("build %.2f s" % (time.perf_counter() - t0), flush=True)
Qd = gen.rows(bench.QUERY_SEED, 0, 4096)
Q = Qd.cpu().numpy()
ok = torch.empty((1024, k), dtype=torch.int64, device=dev)
od = torch.empty((1024, k), dtype=torch.float32, device=dev)
oc = torch.empty(1024, dtype=torch.int32, device=dev)
ref = None
for mode, name in ((0, "workgroups"), (2, "solo")):
    idx.set_search_solo(mode)
    for i in range(32):
        idx.search(Q[i], k, ef)
    t0 = time.perf_counter()
    n = 2000
    for i in range(n):
        idx.search(Q[i % 4096], k, ef)
    dt = time.perf_counter() - t0
    kms = 0.0
    for i in range(256):
        idx.search(Q[i], k, ef)
        kms += idx.timing()["search_kernel_ms"]
    st = idx.last_search_stats()
    print("%-10s single query: %.1f us per call (%.0f q/s), kernel %.1f us" % (name, dt / n * 1e6, n / dt, kms / 256 * 1e3), flush=True)
    got = idx.search_batch(Q[:512], k, ef)
    stq = idx.last_search_stats()
    if ref is None:
        ref = got
        print("           %.1f distances, %.1f expansions per query" % (stq[0] / 512, stq[1] / 512))
    else:
        assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32)), "shapes disagree"
    for nq in (4, 16, 32, 64, 128, 256, 1024):
        for _ in range(3):
            idx.search_batch_device(Qd.data_ptr(), nq, k, ef, ok.data_ptr(), od.data_ptr(), oc.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps, kms = 20, 0.0
        for r in range(reps):
            idx.search_batch_device(Qd[(r * nq) % 2048:].data_ptr(), nq, k, ef, ok.data_ptr(), od.data_ptr(), oc.data_ptr())
            kms += idx.timing()["search_kernel_ms"]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%-10s batch %4d: %.1f us per launch (kernel %.1f us) = %.0f q/s" % (name, nq, dt * 1e6, kms / reps * 1e3, nq / dt), flush=True)
