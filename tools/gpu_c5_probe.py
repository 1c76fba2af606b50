"""BASELINE configs[4] on ONE shard (not a pytest module): FLOAT[1536] ip top-100 — bulk build, batched search, then
insert 1 % new rows, delete 1 % random rows, compact, re-measuring recall and throughput after every step.

    python tools/gpu_c5_probe.py [rows=12500000] [M=32] [ef_construction=128]

12.5M rows is the per-GPU share of the 100M-row configuration (76.8 GB of vectors per GPU).
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

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32
efc = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dim, metric, B, k = 1536, "ip", 1024, 100
extra = rows // 100
pkg = load_package()
dev = torch.device("cuda", 0)
gen = bench.Mixture(rows + extra, dim, True, dev)
idx = pkg.GpuIndex(dim, metric, M, 2 * M, efc)
idx.reserve(rows + extra)
CH = bench.CHUNK


def stage(first, n, key0):
    pos = 0
    while pos < n:
        m = min(CH, n - pos)
        x = gen.rows(bench.DATA_SEED, (first + pos) // CH + 100_000 * (key0 > 0), m)
        ids = torch.arange(key0 + first + pos, key0 + first + pos + m, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        idx.stage_device(ids.data_ptr(), x.data_ptr(), m)
        pos += m


t0 = time.perf_counter()
stage(0, rows, 0)
torch.cuda.synchronize()
t1 = time.perf_counter()
idx.build_finalize()
torch.cuda.synchronize()
t2 = time.perf_counter()
tm = idx.timing(reset=True)
print("build %d x %d %s M=%d efc=%d: stage %.1f s, link %.1f s = %.0f rows/s (phase A %.1f s, phase B %.1f s, %d batches); "
      "device memory %.1f GB" % (rows, dim, metric, M, efc, t1 - t0, t2 - t1, rows / (t2 - t1), tm["build_phase_a_ms"] / 1e3,
                                 tm["build_phase_b_ms"] / 1e3, tm["build_batches"], idx.memory_usage() / 1e9))

Q = gen.rows(bench.QUERY_SEED, 0, B)
ok = torch.empty((B, k), dtype=torch.int64, device=dev)
od = torch.empty((B, k), dtype=torch.float32, device=dev)
oc = torch.empty(B, dtype=torch.int32, device=dev)
tk = torch.empty((B, k), dtype=torch.int64, device=dev)


def measure(what, efs=(128, 192, 256)):
    t = time.perf_counter()
    idx.search_batch_device(Q.data_ptr(), B, k, 0, tk.data_ptr(), od.data_ptr(), oc.data_ptr(), exact=True)
    torch.cuda.synchronize()
    t_exact = time.perf_counter() - t
    line = []
    for ef in efs:
        idx.search_batch_device(Q.data_ptr(), B, k, ef, ok.data_ptr(), od.data_ptr(), oc.data_ptr())
        torch.cuda.synchronize()
        ms = idx.timing()["search_kernel_ms"]
        st = idx.last_search_stats()
        rec = bench.recall_at_k(ok, tk)
        line.append("ef %d: recall@%d %.4f, %.2f ms/batch = %.0f q/s, %.0f dists/query" % (ef, k, rec, ms, B / ms * 1e3,
                                                                                          st[0] / B))
    print("%s (size %d, nodes %d; exact ground truth %.2f s): %s" % (what, idx.size(), idx.nodes(), t_exact, " | ".join(line)))
    return tk.clone()


measure("after bulk build")

# delete 1 % random rows (HNSWIndex::Delete), search must never return them
g = torch.Generator(device="cpu").manual_seed(1234)
dead = torch.randperm(rows, generator=g)[:extra].numpy().astype(np.int64)
t = time.perf_counter()
removed = idx.remove(dead)
print("removed %d rows in %.2f s" % (removed, time.perf_counter() - t))
truth = measure("after deleting 1 %")
assert not np.isin(truth.cpu().numpy(), dead).any(), "a deleted row was returned"

# insert 1 % new rows: the first ones take over tombstoned slots through the update() path — only as many as usearch's
# free ring still remembers after it wrapped (at most 64: quirk Q11, replicated) — the rest are appended
t = time.perf_counter()
stage(rows, extra, 0)
idx.build_finalize()
torch.cuda.synchronize()
tm = idx.timing(reset=True)
print("inserted %d rows in %.2f s (%d batches): nodes %d (slots reused: %d)" % (extra, time.perf_counter() - t,
                                                                                tm["build_batches"], idx.nodes(),
                                                                                rows + extra - idx.nodes()))
measure("after inserting 1 %")

dead2 = rows + torch.randperm(extra, generator=g)[:extra // 2].numpy().astype(np.int64)
print("removed %d of the new rows" % idx.remove(dead2))
t = time.perf_counter()
idx.compact()
print("compact: %.2f s" % (time.perf_counter() - t))
measure("after compact")
