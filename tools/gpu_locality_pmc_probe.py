"""Round 3, locality with counters: the SAME 16-batch k_search launches over (A) the index in insertion order, (B) after
vss_compact's (level, cluster) reordering, (C) reordered AND the queries handed out in the order of the row they land next to
(neighbouring queries run at the same time: what per-XCD query buckets would give, without touching the kernel).
Run under rocprofv3 --kernel-trace --pmc ...; tools/gpu_locality_pmc.sh groups the k_search dispatches by launch number:
  launches 0-3 = A (1 warm-up + 3), 4-7 = B, 8 = the top-1 pre-search of C, 9-12 = C.
    python tools/gpu_locality_pmc_probe.py [rows]"""
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
dim, metric, M, efc, ef, k, B, G = 768, "cosine", 32, 256, 96, 10, 1024, 16
dev = torch.device("cuda", 0)
pkg = load_package()
gen = bench.Mixture(rows, dim, True, dev)
idx = pkg.GpuIndex(dim, metric, M, 2 * M, efc, ef)
idx.reserve(rows)
for c in range(0, rows, bench.CHUNK):
    m = min(bench.CHUNK, rows - c)
    x = gen.rows(bench.DATA_SEED, c // bench.CHUNK, m)
    ids = torch.arange(c, c + m, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    idx.stage_device(ids.data_ptr(), x.data_ptr(), m)
    del x, ids
idx.build_finalize()
Qall = torch.cat([gen.rows(bench.QUERY_SEED, i, B) for i in range(G)])  # 16384 queries
outs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
         torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(G)]


def launch(Q, kk=k, e=ef):
    idx.search_multi_begin(0, [Q[i * B:(i + 1) * B].data_ptr() for i in range(G)], B, kk, e, [o[0].data_ptr() for o in outs],
                           [o[1].data_ptr() for o in outs], [o[2].data_ptr() for o in outs])
    idx.search_end(0)
    st = idx.last_search_stats()
    return idx.timing()["search_kernel_ms"], int(st[0]), int(st[1])


def measure(tag, Q):
    launch(Q)
    ms, nd, ne = 0.0, 0, 0
    for _ in range(3):
        a, b, c = launch(Q)
        ms, nd, ne = ms + a, nd + b, ne + c
    by = nd * (4 * dim + 4) + ne * (4 + 8 * M)
    print("%-44s %.2f ms per launch, algorithmic %.2f GB per launch = %.0f GB/s" % (tag, ms / 3, by / 3 / 1e9, by / (ms / 1e3) / 1e9), flush=True)


measure("A insertion order", Qall)
t0 = time.perf_counter()
idx.compact(True)
print("vss_compact %.2f s" % (time.perf_counter() - t0), flush=True)
measure("B (level, cluster) order", Qall)
launch(Qall, 1, 16)  # launch 8 (keeps the launch numbering documented above; its answers are not used)
# spatial key of a query: the mixture component it was drawn from (bench.Mixture.rows draws the assignment first, from the
# same seeded generator) — queries of one component land in the same region of the graph
comp = []
for i in range(G):
    g = torch.Generator(device=dev).manual_seed(bench.QUERY_SEED + 7919 * i)
    comp.append(torch.randint(0, gen.k, (B,), generator=g, device=dev))
order = torch.argsort(torch.cat(comp), stable=True).cpu().numpy()
Qsorted = Qall[torch.from_numpy(order).to(dev)].contiguous()
measure("C reordered + queries sorted by mixture component", Qsorted)
