"""Per-phase shader-clock breakdown of k_search (needs the -DVSS_PHASE_TIMERS debug build: libvssgpu_prof.so).

    VSS_LIBRARY=duckdb-vss_amd/libvssgpu_prof.so python tools/gpu_phase_probe.py [rows] [dim] [metric] [M] [efc] [ef,ef,..]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
metric = sys.argv[3] if len(sys.argv) > 3 else "cosine"
M = int(sys.argv[4]) if len(sys.argv) > 4 else 32
efc = int(sys.argv[5]) if len(sys.argv) > 5 else 256
efs = [int(e) for e in sys.argv[6].split(",")] if len(sys.argv) > 6 else [96, 320]
pkg = load_package()
dev = torch.device("cuda", 0)
gen = bench.Mixture(rows, dim, metric != "l2sq", dev)
idx = pkg.GpuIndex(dim, metric, M, 2 * M, efc)
idx.reserve(rows)
x = gen.rows(bench.DATA_SEED, 0, rows)
ids = torch.arange(rows, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
idx.stage_device(ids.data_ptr(), x.data_ptr(), rows)
idx.build_finalize()
print("built", rows, idx.timing(reset=True))
k = 10
for B in (64, 1024):
    q = gen.rows(bench.QUERY_SEED, 0, B)
    ok = torch.empty((B, k), dtype=torch.int64, device=dev)
    od = torch.empty((B, k), dtype=torch.float32, device=dev)
    oc = torch.empty(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for ef in efs:
        for _ in range(2):
            idx.search_batch_device(q.data_ptr(), B, k, ef, ok.data_ptr(), od.data_ptr(), oc.data_ptr())
        ms = idx.timing()["search_kernel_ms"]
        st = idx.last_search_stats()
        ticks = np.zeros((B, 12), dtype=np.uint64)
        rc = idx.lib.vss_debug_phase_ticks(idx.h, ticks.ctypes.data, B)
        assert rc == 0
        t = ticks.astype(np.float64)
        mean = t.mean(0)
        print("B=%d ef=%d kernel %.3f ms; per query: dists %.0f expansions %.0f" % (B, ef, ms, st[0] / B, st[1] / B))
        print("   mean ticks: pick %.0f gather %.0f dist %.0f accept %.0f descend %.0f total %.0f  (max total %.0f)" % (
            *mean[:6], t[:, 5].max()))
        ne = st[1] / B
        print("   dist phase per expansion (walker): post + look-ahead %.0f, waiting for the scoring waves %.0f" % (
            mean[7] / ne, mean[9] / ne))
        idx.search_batch(q.cpu().numpy(), k, ef)  # host-pointer call: keeps the per-query counters
        qs = idx.last_query_stats(B).astype(np.float64)
        order = np.argsort(t[:, 5])
        pct = lambda a, p: float(np.percentile(a, p))
        print("   total ticks p50 %.0f p90 %.0f p99 %.0f max %.0f ; expansions p50 %.0f p90 %.0f p99 %.0f max %.0f" % (
            pct(t[:, 5], 50), pct(t[:, 5], 90), pct(t[:, 5], 99), t[:, 5].max(),
            pct(qs[:, 1], 50), pct(qs[:, 1], 90), pct(qs[:, 1], 99), qs[:, 1].max()))
        slow = order[-8:]
        print("   8 slowest: expansions", qs[slow, 1].astype(int).tolist(), "dists", qs[slow, 0].astype(int).tolist())
        print("   8 slowest ticks/expansion", (t[slow, 5] / qs[slow, 1]).astype(int).tolist(),
              " mean over all", int((t[:, 5] / qs[:, 1]).mean()))
        print("   per expansion: pick %.0f gather %.0f dist %.0f accept %.0f ; ticks/ms of longest query: %.0f" % (
            mean[0] / (st[1] / B), mean[1] / (st[1] / B), mean[2] / (st[1] / B), mean[3] / (st[1] / B), t[:, 5].max() / ms))

