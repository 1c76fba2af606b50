"""Where an expansion's cycles go in the two engine shapes (needs the -DVSS_PHASE_TIMERS build: libvssgpu_prof.so).
    VSS_LIBRARY=duckdb-vss_amd/libvssgpu_prof.so python tools/gpu_solo_phase_probe.py [rows] [dim] [metric] [M] [efc] [ef]"""
import os
import sys

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
idx.build_finalize()
k = 10
Qall = gen.rows(bench.QUERY_SEED, 0, 1024)
for mode, name in ((0, "workgroups"), (2, "solo")):
    idx.set_search_solo(mode)
    for B in (1, 64):
        q = Qall[:B].contiguous()
        ok = torch.empty((B, k), dtype=torch.int64, device=dev)
        od = torch.empty((B, k), dtype=torch.float32, device=dev)
        oc = torch.empty(B, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        # every launch on queries the caches have not seen (PROBE_DISTINCT launches, default 16; 1 = round 3's first probes,
        # which repeated one launch three times and therefore measured warm caches)
        reps = int(os.environ.get("PROBE_DISTINCT", "16"))
        ms, t, nd, ne = 0.0, np.zeros(12), 0.0, 0.0
        for r in range(reps):
            q = Qall[(r * B) % (1024 - B + 1):][:B].contiguous()
            idx.search_batch_device(q.data_ptr(), B, k, ef, ok.data_ptr(), od.data_ptr(), oc.data_ptr())
            ms += idx.timing()["search_kernel_ms"] / reps
            st = idx.last_search_stats()
            ticks = np.zeros((B, 12), dtype=np.uint64)
            assert idx.lib.vss_debug_phase_ticks(idx.h, ticks.ctypes.data, B) == 0
            t += ticks.astype(np.float64).mean(0) / reps
            nd += st[0] / B / reps
            ne += st[1] / B / reps
        st = (nd * B, ne * B)
        print("%-10s B=%4d kernel %.1f us; %.0f dists %.1f expansions per query; ticks per expansion: pick %.0f gather %.0f "
              "dist %.0f accept %.0f | descend (whole) %.0f total per query %.0f -> %.0f per expansion; ticks per us of the kernel: %.0f"
              "; list-cache hits %.2f, lists touched %.2f per expansion"
              % (name, B, ms * 1e3, st[0] / B, ne, t[0] / ne, t[1] / ne, t[2] / ne, t[3] / ne, t[4], t[5], t[5] / ne,
                 ticks[:, 5].max() / (ms * 1e3), t[7] / ne, t[8] / ne), flush=True)
