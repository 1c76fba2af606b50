"""Round 4: the workgroup engine with and without crews (the last walker of a workgroup hands its rows to the scoring waves
behind two barriers instead of through the mailboxes) — launch latency over batch sizes on queries no launch has seen, the
one-query host-pointer probe, one- and ten-batch launches, identical answers.  With the -DVSS_PHASE_TIMERS build
(VSS_LIBRARY=.../libvssgpu_prof.so) also the shader-clock ticks per phase of an expansion.
    python tools/gpu_crew_probe.py [rows] [dim] [metric] [M] [efc] [ef]"""
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

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
metric = sys.argv[3] if len(sys.argv) > 3 else "cosine"
M = int(sys.argv[4]) if len(sys.argv) > 4 else 32
efc = int(sys.argv[5]) if len(sys.argv) > 5 else 256
ef = int(sys.argv[6]) if len(sys.argv) > 6 else 80
prof = "prof" in os.environ.get("VSS_LIBRARY", "")
pkg = load_package()
dev = torch.device("cuda", 0)
gen = bench.Mixture(rows, dim, metric != "l2sq", dev)
idx = pkg.GpuIndex(dim, metric, M, 2 * M, efc)
idx.reserve(rows)
t0 = time.perf_counter()
for c in range(0, rows, bench.CHUNK):
    m = min(bench.CHUNK, rows - c)
    x = gen.rows(bench.DATA_SEED, c // bench.CHUNK, m)
    ids = torch.arange(c, c + m, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    idx.stage_device(ids.data_ptr(), x.data_ptr(), m)
idx.build_finalize()
torch.cuda.synchronize()
print("built %d x %d %s M %d efc %d in %.1f s; ef %d" % (rows, dim, metric, M, efc, time.perf_counter() - t0, ef), flush=True)
idx.set_search_solo(0)  # the workgroup engine for every launch
k, NQ = 10, 16384
Qall = torch.cat([gen.rows(bench.QUERY_SEED, i, 1024) for i in range(NQ // 1024)])
# (name, vss_set_search_crew mode, pipelined): 17 = crews without refinements, 21 = + the walker's SIMD spared, 25 = + no list
# requests by the walker, 29 = both; the default is 17
shapes = (("round 3", 0, False), ("crews+pipe plain", 17, True), ("+spare simd", 21, True), ("+no requests", 25, True),
          ("+both", 29, True))


def phase_line(B):
    ticks = np.zeros((B, 12), dtype=np.uint64)
    assert idx.lib.vss_debug_phase_ticks(idx.h, ticks.ctypes.data, B) == 0
    st = idx.last_search_stats()
    ne = max(1.0, float(st[1]) / B)
    t = ticks.astype(np.float64).mean(0)
    return ("ticks/expansion: pick %.0f gather %.0f dist %.0f accept %.0f | walker: hand-over %.0f look-ahead %.0f waiting for scores %.0f | "
            "first scoring wave: prologue %.0f rows+arithmetic %.0f second barrier %.0f | descend %.0f total/query %.0f "
            "= %.0f per expansion, %.1f expansions" % (
                t[0] / ne, t[1] / ne, t[2] / ne, t[3] / ne, t[11] / ne, t[7] / ne, t[9] / ne, t[6] / ne, t[8] / ne, t[10] / ne,
                t[4], t[5], (t[5] - t[4]) / ne, ne))


for B in (1, 8, 64, 204, 256, 1024):
    reps = max(4, min(32, NQ // (len(shapes) * B)))
    line, answers = [], {}
    for si, (name, crew, pipe) in enumerate(shapes):
        idx.set_search_crew(crew)
        idx.set_search_pipelined(pipe)
        ok = torch.empty((reps, B, k), dtype=torch.int64, device=dev)
        od = torch.empty((reps, B, k), dtype=torch.float32, device=dev)
        oc = torch.empty((reps, B), dtype=torch.int32, device=dev)
        kms, extra = 0.0, ""
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):  # every shape gets fresh queries: never one it (or the caches) saw in the previous launch
            q = Qall[((r * len(shapes) + si) * B) % (NQ - B):][:B]
            idx.search_batch_device(q.data_ptr(), B, k, ef, ok[r].data_ptr(), od[r].data_ptr(), oc[r].data_ptr())
            kms += idx.timing()["search_kernel_ms"]
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        if prof and B in (1, 204):
            extra = "\n      %s: %s" % (name, phase_line(B))
        line.append("%s %.1f us (kernel %.1f)%s" % (name, wall * 1e6, kms / reps * 1e3, extra))
    q = Qall[:B]
    for name, crew, pipe in shapes:
        idx.set_search_crew(crew)
        idx.set_search_pipelined(pipe)
        ok1 = torch.empty((B, k), dtype=torch.int64, device=dev)
        od1 = torch.empty((B, k), dtype=torch.float32, device=dev)
        oc1 = torch.empty(B, dtype=torch.int32, device=dev)
        idx.search_batch_device(q.data_ptr(), B, k, ef, ok1.data_ptr(), od1.data_ptr(), oc1.data_ptr())
        torch.cuda.synchronize()
        answers[name] = (ok1.cpu().numpy(), od1.cpu().numpy().view(np.uint32), idx.last_search_stats()[:2].copy())
    a = answers["round 3"]
    same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) for b in answers.values())
    print("B=%4d  " % B + " | ".join(line) + " | identical answers and counters: %s" % same, flush=True)

# launches of 1 and 10 batches of 1024 queries, one launch at a time (the 1x1 regime and the timed launch shape of the bench)
Bq = 1024
for G in (1, 10):
    qs = [Qall[i * Bq:(i + 1) * Bq] for i in range(G)]
    outs = [(torch.empty((Bq, k), dtype=torch.int64, device=dev), torch.empty((Bq, k), dtype=torch.float32, device=dev),
             torch.empty(Bq, dtype=torch.int32, device=dev)) for _ in range(G)]
    for name, crew, pipe in shapes:
        idx.set_search_crew(crew)
        idx.set_search_pipelined(pipe)
        best, total, n = 1e9, 0.0, 6
        for r in range(n + 1):
            idx.search_multi_begin(0, [q.data_ptr() for q in qs], Bq, k, ef, [o[0].data_ptr() for o in outs],
                                   [o[1].data_ptr() for o in outs], [o[2].data_ptr() for o in outs])
            idx.search_end(0)
            ms = idx.timing()["search_kernel_ms"]
            if r:
                best, total = min(best, ms), total + ms
        st = idx.last_search_stats()
        gb = (float(st[0]) * (4 * dim + 4) + float(st[1]) * (4 + 8 * M)) / 1e9
        print("%2d x 1024 queries per launch, %-18s: kernel %.3f ms avg, %.3f best -> %.0f GB/s = %.3f of 8 TB/s (avg)" % (
            G, name, total / n, best, gb / (total / n / 1e3), gb / (total / n / 1e3) / 8000), flush=True)

# the one-query probe of HNSW_INDEX_SCAN through host pointers (vss_search: pinned block, flag wait)
Qh = Qall[:4096].cpu().numpy()
for name, crew, pipe in shapes:
    idx.set_search_crew(crew)
    idx.set_search_pipelined(pipe)
    for i in range(32):
        idx.search(Qh[i], k, ef)
    t0 = time.perf_counter()
    n = 600
    for i in range(n):
        idx.search(Qh[32 + i], k, ef)
    print("vss_search, one query per call, %-18s: %.1f us per call" % (name, (time.perf_counter() - t0) / n * 1e6), flush=True)
    chunk = 204
    for i in range(3):
        idx.search_batch(Qh[i * chunk:(i + 1) * chunk], k, ef)
    t0 = time.perf_counter()
    for i in range(3, 15):
        idx.search_batch(Qh[i * chunk:(i + 1) * chunk], k, ef)
    print("vss_search_batch, %d queries per call (HNSW_INDEX_JOIN chunk), %-18s: %.1f us per call" % (
        chunk, name, (time.perf_counter() - t0) / 12 * 1e6), flush=True)
