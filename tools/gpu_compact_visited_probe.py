"""Round 4 (VSS_VISITED_COMPACT=0/1, on by default): the compact exact visited set (16-bit cells: tag +
displacement) that keeps the sets of searches with limits of 257-512 in LDS.
  part A  exactness: the engine with the knob on against the engine with the knob off (whose answers the parity suite holds
          to the oracle) — row ids, distance bits, result counts and both per-query work counters — over three metrics, plain /
          tombstoned / slot-reusing graphs, a predicate, one-query launches (the roomy table) and a table forced so small
          that displacements and counts overflow (the re-run path);
  part B  what it buys: launches of 10 x 1024 queries at ef 480 / top-100 over a configs[4]-like shard (ip, 1536 dims).
    python tools/gpu_compact_visited_probe.py [rows_b] [dim_b]"""
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

pkg = load_package()
dev = torch.device("cuda", 0)
failures = []


def knob(ix, on, lds_max=None):  # (round 5: a setter of the index; the environment is read once, in vss_create)
    ix.set_search_visited_set(bool(on), lds_max or 0)


def answers(ix, Q, k, ef, allowed=None):
    if len(Q) == 1:
        keys = ix.search(Q[0], k, ef)
        return (keys.copy(),)
    if allowed is not None:
        keys, d, cnt = ix.search_batch_filtered(Q, k, ef, allowed, 40_000)
    else:
        keys, d, cnt = ix.search_batch(Q, k, ef)
    st = ix.last_search_stats()
    return keys, d.view(np.uint32), cnt, ix.last_query_stats(len(Q)), int(st[3])


def compare(tag, ix, Q, k, ef, allowed=None, lds_max=None, want_reruns=False):
    knob(ix, False)
    a = answers(ix, Q, k, ef, allowed)
    knob(ix, True, lds_max)
    b = answers(ix, Q, k, ef, allowed)
    knob(ix, False)
    same = all(np.array_equal(x, y) for x, y in zip(a[:4], b[:4]))
    reruns = (a[4], b[4]) if len(a) > 4 else None
    ok = same and (not want_reruns or b[4] > a[4])
    print("  %-72s %s  re-run queries off / on: %s" % (tag, "identical" if same else "DIFFERENT", reruns), flush=True)
    if not ok:
        failures.append(tag)


# ------------------------------------------------------------------------------------------------ part A
rng = np.random.default_rng(7)
for metric, M in (() if os.environ.get("PROBE_PART_B_ONLY") else (("l2sq", 16), ("cosine", 16), ("ip", 32))):
    n, dim = 40_000, 48
    X = rng.standard_normal((n, dim), dtype=np.float32)
    centres = rng.standard_normal((64, dim), dtype=np.float32) * 3
    X += centres[rng.integers(0, 64, n)]
    if metric == "ip":
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    ix = pkg.GpuIndex(dim, metric, M, 2 * M, 96)
    ix.reserve(n + 4000)
    ix.stage(np.arange(n, dtype=np.int64), X)
    ix.build_finalize()
    ix.set_search_solo(0)  # every launch through the workgroup engine (the solo / team shapes never take the compact form)
    Q = X[rng.integers(0, n, 1024)] + 0.05 * rng.standard_normal((1024, dim), dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    print("part A: %s, %d x %d, M %d" % (metric, n, dim, M), flush=True)
    for nq, k, ef in ((1024, 10, 300), (1024, 100, 480), (200, 100, 257), (1024, 10, 512)):
        compare("plain graph: %d queries, k %d, ef %d" % (nq, k, ef), ix, Q[:nq], k, ef)
    for i in range(4):
        compare("one query per call (roomy table): query %d, k 10, ef 400" % i, ix, Q[i:i + 1], 10, 400)
    compare("table forced to 2^11 cells (overflows: re-run with the plain table)", ix, Q, 100, 480, lds_max=10, want_reruns=True)
    bits = np.zeros((n + 63) // 64 * 64, dtype=np.uint8)
    bits[:n] = rng.random(n) < 0.3
    allowed = np.packbits(bits, bitorder="little").view(np.uint64).copy()
    compare("predicate admits 30 %: 1024 queries, k 10, ef 480", ix, Q, 10, 480, allowed=allowed)
    gone = rng.choice(n, 1200, replace=False).astype(np.int64)
    ix.remove(gone)
    compare("3 % tombstones: 1024 queries, k 100, ef 480", ix, Q, 100, 480)
    compare("3 % tombstones: 1024 queries, k 10, ef 300", ix, Q, 10, 300)
    Y = X[rng.integers(0, n, 1200)] + 0.1 * rng.standard_normal((1200, dim), dtype=np.float32)
    ix.add(np.arange(n, n + 1200, dtype=np.int64), np.ascontiguousarray(Y, dtype=np.float32))
    compare("slots re-used (lists may name a slot twice): 1024 queries, k 100, ef 480", ix, Q, 100, 480)
    compare("slots re-used: 1024 queries, k 10, ef 257", ix, Q, 10, 257)
    ix.close()
if not os.environ.get("PROBE_PART_B_ONLY"):
    print("part A:", "ALL IDENTICAL" if not failures else "FAILED: %s" % failures, flush=True)

# ------------------------------------------------------------------------------------------------ part B
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
metric, M, efc, k, ef, B, G = os.environ.get("PROBE_METRIC", "ip"), int(os.environ.get("PROBE_M", 16)), 128, 100, 480, 1024, 10
gen = bench.Mixture(rows, dim, metric != "l2sq", dev)
ix = pkg.GpuIndex(dim, metric, M, 2 * M, efc)
ix.reserve(rows)
for c in range(0, rows, bench.CHUNK):
    m = min(bench.CHUNK, rows - c)
    x = gen.rows(bench.DATA_SEED, c // bench.CHUNK, m)
    ids = torch.arange(c, c + m, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ix.stage_device(ids.data_ptr(), x.data_ptr(), m)
    del x, ids
t0 = time.perf_counter()
ix.build_finalize()
torch.cuda.synchronize()
print("part B: built %d x %d %s M %d efc %d in %.1f s" % (rows, dim, metric, M, efc, time.perf_counter() - t0), flush=True)
Q = [gen.rows(bench.QUERY_SEED, i, B) for i in range(G)]
outs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
         torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(G)]
torch.cuda.synchronize()
ref = None
for e in (ef, 320):
    for name, on in (("32-bit sets in HBM", False), ("compact sets in LDS", True), ("32-bit sets in HBM (again)", False),
                     ("compact sets in LDS (again)", True)):
        knob(ix, on)
        ms_k, ms_w = [], []
        for r in range(3):
            torch.cuda.synchronize()
            tw = time.perf_counter()
            ix.search_multi_begin(0, [q.data_ptr() for q in Q], B, k, e, [o[0].data_ptr() for o in outs],
                                  [o[1].data_ptr() for o in outs], [o[2].data_ptr() for o in outs])
            ix.search_end(0)
            ms_w.append((time.perf_counter() - tw) * 1e3)
            ms_k.append(ix.timing()["search_kernel_ms"])
        st = ix.last_search_stats()
        gb = (float(st[0]) * (4 * dim + 4) + float(st[1]) * (4 + 8 * M)) / 1e9
        ans = (outs[0][0].cpu().numpy().copy(), outs[0][1].cpu().numpy().view(np.uint32).copy(), int(st[0]), int(st[1]))
        if ref is None or ref[0] != e:
            ref = (e, ans)
        same = all(np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b for a, b in zip(ref[1], ans))
        if not same:
            failures.append("part B ef %d %s" % (e, name))
        kms, wms = min(ms_k[1:]), min(ms_w[1:])
        print("  ef %3d  %-34s %d x %d queries: kernels %.2f ms (wall incl. re-runs %.2f) -> %.0f queries/s over wall, %.3f of 8 TB/s "
              "over wall; distances/query %.0f; re-run queries %d; identical answers %s" % (
                  e, name, G, B, kms, wms, G * B / wms * 1e3, gb / (wms / 1e3) / 8000, float(st[0]) / (G * B), int(st[3]), same), flush=True)
knob(ix, False)
print("RESULT:", "ok" if not failures else "FAILED %s" % failures, flush=True)
sys.exit(1 if failures else 0)
