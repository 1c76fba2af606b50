"""Round 4: is the workgroup engine short of WALKERS when expansions are thin (large ef_search: 768 dims at ef 128 runs at
0.67 of the HBM peak against 0.79 at ef 64 whatever the table size; 12.5M x 1536 at ef 384-480 at 0.57-0.63)?  Launches of
10 x 1024 queries with 4, 6 and 8 walkers per workgroup (the rest of the 16 waves score), visited sets in LDS or in HBM.
    python tools/gpu_walkers_probe.py rows dim metric M efc k ef[,ef...]"""
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

rows, dim, metric, M, efc, k = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
efs = [int(e) for e in sys.argv[7].split(",")]
B, G = 1024, 10
pkg = load_package()
dev = torch.device("cuda", 0)
gen = bench.Mixture(rows, dim, metric != "l2sq", dev)
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
print("built %d x %d %s M %d efc %d in %.1f s" % (rows, dim, metric, M, efc, time.perf_counter() - t0), flush=True)
Q = [gen.rows(bench.QUERY_SEED, i, B) for i in range(G)]
outs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
         torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(G)]
torch.cuda.synchronize()
ref = {}
for ef in efs:
    variants = (("4 walkers (default)", 0, None, None), ("4 walkers, visited sets in HBM", 4, 10, None), ("6 walkers, visited sets in HBM", 6, 10, None),
                ("8 walkers, visited sets in HBM", 8, 10, None))
    if os.environ.get("PROBE_TABLES"):  # smaller visited sets that stay in LDS (overflowing queries are re-run), wall clock incl. retries
        variants = (("default sizing (64 cells per limit)", 0, None, None), ("LDS tables up to 64 KiB (fewer walkers)", 0, 14, None),
                    ("32 cells per limit", 0, None, 32), ("16 cells per limit", 0, None, 16), ("8 cells per limit", 0, None, 8))
    for name, walkers, hash_lds, per_limit in variants:
        idx.set_search_visited_set(True, hash_lds or 0, per_limit or 0)
        idx.set_search_params(16, walkers)
        ms_all = []
        for r in range(3):
            torch.cuda.synchronize()
            tw = time.perf_counter()
            idx.search_multi_begin(0, [q.data_ptr() for q in Q], B, k, ef, [o[0].data_ptr() for o in outs],
                                   [o[1].data_ptr() for o in outs], [o[2].data_ptr() for o in outs])
            idx.search_end(0)
            ms_all.append((time.perf_counter() - tw) * 1e3 if os.environ.get("PROBE_TABLES") else idx.timing()["search_kernel_ms"])
        st = idx.last_search_stats()
        gb = (float(st[0]) * (4 * dim + 4) + float(st[1]) * (4 + 8 * M)) / 1e9
        ms = min(ms_all[1:])
        ans = (outs[0][0].cpu().numpy().copy(), outs[0][1].cpu().numpy().view(np.uint32).copy(), int(st[0]), int(st[1]))
        ref.setdefault(ef, ans)
        same = all(np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b for a, b in zip(ref[ef], ans))
        print("ef %3d  %-40s launch of %d x %d queries %.2f ms -> %.0f queries/s, %.0f GB/s = %.3f of 8 TB/s; re-run queries %d; identical answers %s" % (
            ef, name, G, B, ms, G * B / ms * 1e3, gb / (ms / 1e3), gb / (ms / 1e3) / 8000, int(st[3]), same), flush=True)
