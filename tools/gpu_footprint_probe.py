"""Round 4: does `k_search`'s share of the HBM peak depend on the SIZE of the table rather than on the width of its rows?
(A 12.5M x 1536 shard — 77 GB — runs at 0.57-0.65 whatever the kernel shape, a 3M x 1536 one — 18 GB — at 0.80, 10M x 768 —
31 GB — at 0.77-0.81.)  Same dimension (768), same options (M 32, ef_construction 128: a cheap graph), same ef: 10M rows
against 25M rows (77 GB), launches of 10 x 1024 queries.
    python tools/gpu_footprint_probe.py [dim=768] rows [rows ...]"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 768
sizes = [int(a) for a in sys.argv[2:]] or [10_000_000, 25_000_000]
metric, M, efc, ef, B, k, G = "cosine", 32, 128, 64, 1024, 10, 10
pkg = load_package()
dev = torch.device("cuda", 0)
for rows in sizes:
    gen = bench.Mixture(rows, dim, True, dev)
    idx = pkg.GpuIndex(dim, metric, M, 2 * M, efc)
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
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    Q = [gen.rows(bench.QUERY_SEED, i, B) for i in range(G)]
    outs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
             torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(G)]
    torch.cuda.synchronize()
    for e in (ef, 2 * ef):
        ms_all = []
        for r in range(4):
            idx.search_multi_begin(0, [q.data_ptr() for q in Q], B, k, e, [o[0].data_ptr() for o in outs],
                                   [o[1].data_ptr() for o in outs], [o[2].data_ptr() for o in outs])
            idx.search_end(0)
            ms_all.append(idx.timing()["search_kernel_ms"])
        st = idx.last_search_stats()
        gb = (float(st[0]) * (4 * dim + 4) + float(st[1]) * (4 + 8 * M)) / 1e9
        ms = min(ms_all[1:])
        print("%9d rows x %d (%.1f GB of vectors; built in %.1f s): ef %3d, launch of %d x %d queries %.2f ms, %.0f distances / %.0f expansions per "
              "query -> %.0f GB/s = %.3f of 8 TB/s" % (rows, dim, rows * dim * 4 / 1e9, build_s, e, G, B, ms, st[0] / (G * B), st[1] / (G * B),
                                                       gb / (ms / 1e3), gb / (ms / 1e3) / 8000), flush=True)
    idx.close()
    del idx, gen, Q, outs
    torch.cuda.empty_cache()
