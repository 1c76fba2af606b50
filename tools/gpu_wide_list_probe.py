"""Round 5: limits of 257-512 (the 8-register candidate list).  Launches of 10 x 1024 queries and of one batch, the same queries
through (a) round 4's shape — 16 waves, plain order — and (b) 12-wave workgroups with the pipelined level search
(vss_set_search_wide_lists); ids, distance bits and both work counters must be identical.
    python tools/gpu_wide_list_probe.py rows dim metric M efc k ef[,ef...]"""
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
PROF = bool(os.environ.get("VSS_LIBRARY"))  # the -DVSS_PHASE_TIMERS build: shader-clock ticks per phase of an expansion, walker's view


def phases(n):
    ticks = np.zeros((n, 12), dtype=np.uint64)
    assert idx.lib.vss_debug_phase_ticks(idx.h, ticks.ctypes.data, n) == 0
    st = idx.last_search_stats()
    ne = max(1.0, float(st[1]) / n)
    t = ticks.astype(np.float64).mean(0)
    return ("\n      ticks per expansion: pick %.0f gather %.0f scores %.0f accept %.0f | hand-over %.0f look-ahead %.0f waiting %.0f | descend %.0f "
            "per query, total %.0f = %.0f per expansion" % (t[0] / ne, t[1] / ne, t[2] / ne, t[3] / ne, t[11] / ne, t[7] / ne, t[9] / ne, t[4],
                                                             t[5], (t[5] - t[4]) / ne))


VARIANTS = [("16 waves, plain order (round 4)", dict(wide=False, retry=False)), ("12 waves, pipelined, host re-runs", dict(wide=True, retry=False)),
            ("12 waves, pipelined, retry in place", dict(wide=True, retry=True))]
if os.environ.get("PROBE_WALKERS"):  # e.g. "5:12,6:12,6:13" = walkers : log2 of the 32-bit words under a walker's visited set
    VARIANTS = VARIANTS[2:]
    for spec in os.environ["PROBE_WALKERS"].split(","):
        w, l2 = (int(t) for t in spec.split(":"))
        VARIANTS.append(("%d walkers, sets of 2^%d cells" % (w, l2 + 1), dict(wide=True, retry=True, walkers=w, lds_log2=l2)))
if os.environ.get("PROBE_EXTRA"):
    VARIANTS += [("12 waves, plain order", dict(wide=True, pipelined=False)), ("16 waves, sets in HBM", dict(wide=False, compact=False))]
bad = 0
for ef in efs:
    ref = {}
    for g in (G, 1):
        for name, v in VARIANTS:
            idx.set_search_wide_lists(v.get("wide", True))
            idx.set_search_pipelined(v.get("pipelined", True))
            idx.set_search_visited_set(v.get("compact", True), v.get("lds_log2", 0), int(os.environ.get("PROBE_PER_LIMIT", "0")), v.get("retry", True))
            if v.get("wide") and not v.get("pipelined", True):
                idx.set_search_params(12, 0)
            else:
                idx.set_search_params(16, v.get("walkers", 0))
            ms_all = []
            for r in range(3):
                torch.cuda.synchronize()
                tw = time.perf_counter()
                idx.search_multi_begin(0, [q.data_ptr() for q in Q[:g]], B, k, ef, [o[0].data_ptr() for o in outs[:g]],
                                       [o[1].data_ptr() for o in outs[:g]], [o[2].data_ptr() for o in outs[:g]])
                idx.search_end(0)
                wall = (time.perf_counter() - tw) * 1e3
                ms_all.append(wall if os.environ.get("PROBE_WALL", "1") == "1" else idx.timing()["search_kernel_ms"])  # wall: incl. re-runs
            st = idx.last_search_stats()
            gb = (float(st[0]) * (4 * dim + 4) + float(st[1]) * (4 + 8 * M)) / 1e9
            ms = min(ms_all[1:])
            ans = (outs[0][0].cpu().numpy().copy(), outs[0][1].cpu().numpy().view(np.uint32).copy(), outs[0][2].cpu().numpy().copy(),
                   outs[g - 1][0].cpu().numpy().copy(), outs[g - 1][1].cpu().numpy().view(np.uint32).copy(),
                   np.array([int(st[0]), int(st[1])]))  # (work counters: the launch's totals)
            same = all(np.array_equal(a, b) for a, b in zip(ref.setdefault(g, ans), ans))
            bad += not same
            print("ef %3d  %-34s %2d x %d queries %7.2f ms -> %7.0f queries/s, %5.0f GB/s = %.3f of 8 TB/s; %.0f distances %.1f expansions "
                  "per query; re-run %d; identical %s%s" % (ef, name, g, B, ms, g * B / ms * 1e3, gb / (ms / 1e3), gb / (ms / 1e3) / 8000,
                                                             float(st[0]) / (g * B), float(st[1]) / (g * B), int(st[3]), same,
                                                             phases(g * B) if PROF else ""), flush=True)
print("DIFFERENCES: %d" % bad)
sys.exit(1 if bad else 0)
