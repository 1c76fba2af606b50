"""Build + batched search at a given dimension (round 3: unrolled instantiations for 512 / 1024 dims against the looping ones).
   [VSS_FORCE_LOOPING=1] python tools/gpu_dim_probe.py rows dim"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

rows, dim = int(sys.argv[1]), int(sys.argv[2])
metric, M, efc, ef, k, B, G = "cosine", 32, 256, 96, 10, 1024, 8
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
t0 = time.perf_counter()
idx.build_finalize()
tb = time.perf_counter() - t0
Q = [gen.rows(bench.QUERY_SEED, i, B) for i in range(G)]
outs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
         torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(G)]
ms = 0.0
for it in range(4):
    idx.search_multi_begin(0, [q.data_ptr() for q in Q], B, k, ef, [o[0].data_ptr() for o in outs], [o[1].data_ptr() for o in outs],
                           [o[2].data_ptr() for o in outs])
    idx.search_end(0)
    if it:
        ms += idx.timing()["search_kernel_ms"]
st = idx.last_search_stats()
by = int(st[0]) * (4 * dim + 4) + int(st[1]) * (4 + 8 * M)
print("%s dim %d rows %d: build %.2f s (%.0f rows/s); %d-batch launch %.2f ms = %.0f queries/s, %.0f GB/s algorithmic" % (
    "looping kernels" if os.environ.get("VSS_FORCE_LOOPING") else "unrolled kernels", dim, rows, tb, rows / tb, G, ms / 3, G * B / (ms / 3e3),
    by / (ms / 3e3) / 1e9))
