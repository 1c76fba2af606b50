"""Round 4: where an expansion's cycles go at limits beyond 256 (the 8-register list's instantiation of k_search: plain order,
one list in flight) against limits of 129-256 (4 registers: pipelined level search) — shader-clock ticks per phase, walker's view.
Needs the -DVSS_PHASE_TIMERS build:  VSS_LIBRARY=$PWD/duckdb-vss_amd/libvssgpu_prof.so python tools/gpu_large_ef_phase_probe.py [rows] [dim]"""
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

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
metric, M, efc, k, B = "ip", 16, 128, 100, 1024
pkg = load_package()
dev = torch.device("cuda", 0)
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
print("built %d x %d %s M %d efc %d in %.1f s" % (rows, dim, metric, M, efc, time.perf_counter() - t0), flush=True)
G = 10
Q = [gen.rows(bench.QUERY_SEED, i, B) for i in range(G)]
outs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
         torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(G)]
torch.cuda.synchronize()


def phases(n):
    ticks = np.zeros((n, 12), dtype=np.uint64)
    assert idx.lib.vss_debug_phase_ticks(idx.h, ticks.ctypes.data, n) == 0
    st = idx.last_search_stats()
    ne = max(1.0, float(st[1]) / n)
    t = ticks.astype(np.float64).mean(0)
    return ("pick %.0f gather %.0f scores %.0f accept %.0f | hand-over %.0f look-ahead %.0f waiting %.0f | descend %.0f, total %.0f = %.0f per "
            "expansion, %.1f expansions, %.0f distances per query" % (
                t[0] / ne, t[1] / ne, t[2] / ne, t[3] / ne, t[11] / ne, t[7] / ne, t[9] / ne, t[4], t[5], (t[5] - t[4]) / ne, ne,
                float(st[0]) / n))


for ef in (192, 256, 320, 480):
    for compact in ((0, 1) if ef > 256 else (0,)):
        idx.set_search_visited_set(bool(compact))
        for g in (1, G):
            ms = []
            for r in range(3):
                idx.search_multi_begin(0, [q.data_ptr() for q in Q[:g]], B, k, ef, [o[0].data_ptr() for o in outs[:g]],
                                       [o[1].data_ptr() for o in outs[:g]], [o[2].data_ptr() for o in outs[:g]])
                idx.search_end(0)
                ms.append(idx.timing()["search_kernel_ms"])
            print("ef %3d compact %d  %2d x %d queries: %.2f ms  ticks/expansion: %s" % (ef, compact, g, B, min(ms[1:]), phases(g * B)), flush=True)
