"""(Needs the 12-wave instantiation of k_search — __launch_bounds__(768), removed after this measurement: commit c531b89 holds it.)  Round 4: k_search at 1536 dimensions — the 16-wave workgroup (2 rows in flight per scoring wave, plain
order beyond 256 entries) against the 12-wave one (170 registers: 4 rows in flight, pipelined with an 8-register list), and
crews / pipelining off, on one shard of configs[4]: launches of 16 x 1024 queries, one at a time, top-100.
    python tools/gpu_wide_row_probe.py [rows=12500000] [ef=448]"""
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

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 448
dim, metric, B, k, M, efc, G = 1536, "ip", 1024, 100, 32, 128, 16
pkg = load_package()
dev = torch.device("cuda", 0)
gen = bench.Mixture(rows + rows // 100, dim, True, dev)
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
print("built %d x %d %s in %.1f s" % (rows, dim, metric, time.perf_counter() - t0), flush=True)
Q = [gen.rows(bench.QUERY_SEED, i, B) for i in range(G)]
outs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
         torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(G)]
torch.cuda.synchronize()
ref = None
for name, waves, crew, pipe in (("16 waves, round 3", 16, 0, False), ("16 waves, crews + pipelined (E <= 4 only)", 16, 17, True),
                                ("12 waves x 4 rows, plain order", 12, 17, False), ("12 waves x 4 rows, crews + pipelined", 12, 17, True)):
    idx.set_search_params(waves, 0)
    idx.set_search_crew(crew)
    idx.set_search_pipelined(pipe)
    for e in (ef, 192):
        ms_all = []
        for r in range(3):
            idx.search_multi_begin(0, [q.data_ptr() for q in Q], B, k, e, [o[0].data_ptr() for o in outs],
                                   [o[1].data_ptr() for o in outs], [o[2].data_ptr() for o in outs])
            idx.search_end(0)
            ms_all.append(idx.timing()["search_kernel_ms"])
        st = idx.last_search_stats()
        gb = (float(st[0]) * (4 * dim + 4) + float(st[1]) * (4 + 8 * M)) / 1e9
        ms = min(ms_all[1:])
        ans = (outs[0][0].cpu().numpy().copy(), outs[0][1].cpu().numpy().view(np.uint32).copy(), int(st[0]), int(st[1]))
        key = (e,)
        if ref is None or key not in ref:
            ref = dict(ref or {})
            ref[key] = ans
        same = all(np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b for a, b in zip(ref[key], ans))
        print("%-44s ef %3d: launch of %d x %d queries %.2f ms -> %.0f queries/s, %.0f GB/s = %.3f of 8 TB/s; identical answers %s" % (
            name, e, G, B, ms, G * B / ms * 1e3, gb / (ms / 1e3), gb / (ms / 1e3) / 8000, same), flush=True)
