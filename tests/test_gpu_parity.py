"""-m gpu: parity of the HIP path (through the C ABI of libvssgpu.so) against the CPU oracle.

Bars (north_star): bit-exact row ids and graph bytes for the integer / indexing work; distances bit-exact against
the oracle's wave-order metric and within 1e-5 relative of the reference-order metric.
The oracle itself is pinned to the reference by tests/test_oracle_golden.py.
"""
import sys

import numpy as np
import pytest

import datagen
import gpu_common as gc
from oracle_lib import CpuIndex, load_oracle, parse_stream

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ------------------------------------------------------------------------------------------------- array_* functions
@pytest.mark.parametrize("dim", [3, 5, 128, 768, 1536])
@pytest.mark.parametrize("fn", ["array_distance", "array_cosine_distance", "array_negative_inner_product"])
def test_array_functions(fn, dim):
    """array_distance = sqrt(sum (a-b)^2), array_cosine_distance = 1 - cos, array_negative_inner_product = -a.b
    (DuckDB core functions named at reference hnsw_index.cpp:659-673; tolerance 1e-5 relative)."""
    rows = 1000
    A = datagen.normals(11 + dim, (rows, dim)).astype(np.float32)
    B = datagen.normals(12 + dim, (rows, dim)).astype(np.float32)
    a64, b64 = A.astype(np.float64), B.astype(np.float64)
    for b_arg, b_ref in ((B, b64), (B[7], b64[7][None, :])):
        got = gc.pkg().distance_batch(fn, A, b_arg)
        if fn == "array_distance":
            ref = np.sqrt(((a64 - b_ref) ** 2).sum(1))
            scale = ref
        elif fn == "array_negative_inner_product":
            ref = -(a64 * b_ref).sum(1)
            scale = np.abs(a64 * b_ref).sum(1)
        else:
            ref = 1 - (a64 * b_ref).sum(1) / np.sqrt((a64 ** 2).sum(1) * (b_ref ** 2).sum(1))
            scale = np.ones_like(ref)
        assert np.all(np.abs(got - ref) <= 1e-5 * np.maximum(scale, 1e-30))


def test_array_distance_readme_values():
    """test/sql/hnsw/hnsw_result.test:23-28: array_distance of the three nearest rows to [1,2,3] = 0.0, 1.0, 1.0."""
    A = np.array([[1, 2, 3], [1, 2, 4], [1, 1, 3]], dtype=np.float32)
    got = gc.pkg().distance_batch("array_distance", A, np.array([1, 2, 3], dtype=np.float32))
    assert got.tolist() == [0.0, 1.0, 1.0]


# ------------------------------------------------------------------------------------------------- search on a given graph
SEARCH_CASES = [(729, 3, "l2sq"), (2000, 16, "l2sq"), (2000, 16, "cosine"), (2000, 20, "ip"), (3000, 128, "cosine"),
                (1500, 768, "l2sq"), (800, 1536, "ip"), (1200, 100, "l2sq")]


@pytest.mark.parametrize("n,dim,metric", SEARCH_CASES)
def test_search_on_reference_built_graph(n, dim, metric):
    """Same graph -> same search: a graph built by the CPU restatement of the reference (sequential add) is loaded
    through the reference's own stream format and searched on the GPU; ids, distance bits and the work counters
    (computed_distances / visited_members) must equal the oracle's."""
    if dim == 3:
        X = datagen.readme_grid()
        Q = np.array([[1, 2, 3], [5, 5, 5], [9, 9, 9], [0.5, 3.25, 7.75]], dtype=np.float32)
    else:
        X, Q = gc.make_data(n, dim, metric, 100 + dim)
    cpu = gc.oracle_index(dim, metric)
    cpu.reserve(len(X))
    cpu.add_many(np.arange(len(X)) * 3 + 1, X)
    gpu = gc.gpu_index(dim, metric)
    gpu.load(cpu.save())
    assert gpu.save() == cpu.save()
    for k, ef in ((10, 64), (3, 16), (100, 0), (10, 200)):
        gk, gd, gcnt = gpu.search_batch(Q, k, ef)
        ck, cd, ccnt, cst = cpu.search_many(Q, k, ef=ef if ef else None)
        assert np.array_equal(gcnt, ccnt)
        assert np.array_equal(gk, ck)
        assert np.array_equal(_bits(gd), _bits(cd))
        assert np.array_equal(gpu.last_query_stats(len(Q)), cst.astype(np.uint32))
    # single-query entry point (HNSW_INDEX_SCAN)
    assert np.array_equal(gpu.search(Q[0], 5), cpu.search(Q[0], 5)[0])


@pytest.mark.parametrize("dim,metric,M", [(128, "l2sq", 16), (768, "cosine", 32), (20, "ip", 8), (512, "ip", 16),
                                          (1024, "l2sq", 16)])
def test_search_kernel_variants_agree(dim, metric, M, monkeypatch):
    """The search engine in every shape — one walker with a single scoring wave, four walkers sharing twelve scoring
    waves, an odd split, the solo shape (k_search_solo), crews on and off, and the per-launch default — takes the reference's decisions in the reference's order: ids,
    distance bits and the work counters (computed_distances, visited_members) are identical for every batch size
    (including batches that make walkers steal queries from the shared counter), and equal to the oracle's."""
    n = 6000
    X, Q = gc.make_data(n, dim, metric, 4100 + dim, nq=700)
    cpu = gc.oracle_index(dim, metric, M, 2 * M, 100)
    cpu.reserve(n)
    cpu.build_batch(np.arange(n), X, 512, 4)
    blob = cpu.save()
    ck, cd, ccnt, cst = cpu.search_many(Q, 10, ef=72)
    variants = {"1 walker + 1 scorer": {"VSS_SEARCH_WAVES": "2", "VSS_SEARCH_WALKERS": "1"},
                "4 walkers + 12 scorers": {"VSS_SEARCH_WAVES": "16", "VSS_SEARCH_WALKERS": "4"},
                "3 walkers + 5 scorers": {"VSS_SEARCH_WAVES": "8", "VSS_SEARCH_WALKERS": "3"},
                # the solo shape (one wave per query scoring its own rows) for every batch size, and never
                # (launches of at most one query per compute unit run it as teams — helper waves score a share of the rows —
                # where the variant exists: narrow rows; "solo, one wave" switches the helpers off)
                "solo": {"VSS_SEARCH_SOLO": "2"}, "engine only": {"VSS_SEARCH_SOLO": "0"},
                "solo, one wave": {"VSS_SEARCH_SOLO": "2", "VSS_SEARCH_TEAM": "0"},
                # round 4, crews: the last walker of a workgroup hands its rows over behind two barriers instead of through the
                # mailboxes — from the start with one walker per workgroup (batches 1, 7, 200), in the drain otherwise;
                # "no crews" is round 3's exchange throughout; look-ahead (which keeps the mailboxes) on top of either
                "engine only, no crews": {"VSS_SEARCH_SOLO": "0", "VSS_SEARCH_CREW": "0"},
                # round 4, the software-pipelined level search (accept in the shadow of the successor's rows) off: plain order
                "engine only, plain order": {"VSS_SEARCH_SOLO": "0", "VSS_SEARCH_PIPELINED": "0"},
                "engine only, round 3": {"VSS_SEARCH_SOLO": "0", "VSS_SEARCH_PIPELINED": "0", "VSS_SEARCH_CREW": "0"},
                "4 walkers + 12 scorers, pipelined": {"VSS_SEARCH_SOLO": "0", "VSS_SEARCH_WAVES": "16", "VSS_SEARCH_WALKERS": "4"},
                "engine only, look-ahead": {"VSS_SEARCH_SOLO": "0"},
                "one walker + 15 scorers": {"VSS_SEARCH_SOLO": "0", "VSS_SEARCH_WALKERS": "1"},
                "2 walkers + 2 scorers": {"VSS_SEARCH_SOLO": "0", "VSS_SEARCH_WAVES": "4", "VSS_SEARCH_WALKERS": "2"},
                "default": {}}
    lookahead = {"1 walker + 1 scorer": 4, "4 walkers + 12 scorers": 4, "engine only, look-ahead": 2, "solo": 2}
    for name, env in variants.items():
        for key in ("VSS_SEARCH_WAVES", "VSS_SEARCH_WALKERS", "VSS_SEARCH_SOLO", "VSS_SEARCH_TEAM", "VSS_SEARCH_CREW",
                    "VSS_SEARCH_PIPELINED"):
            monkeypatch.delenv(key, raising=False)
        for key, value in env.items():
            monkeypatch.setenv(key, value)
        gpu = gc.gpu_index(dim, metric, M, 2 * M, 100)  # the knobs are read when the index is created
        gpu.load(blob)
        # look-ahead off / always on: speculation may never change an id, a distance bit or a work counter
        gpu.set_search_lookahead(lookahead.get(name, 0))
        for batch in (1, 7, 200, 256, 257, 700):  # (host pointers: up to 256 queries take the pinned zero-copy path)
            gk, gd, gcnt = gpu.search_batch(Q[:batch], 10, 72)
            assert np.array_equal(gk, ck[:batch]), (name, batch)
            assert np.array_equal(_bits(gd), _bits(cd[:batch])), (name, batch)
            assert np.array_equal(gcnt, ccnt[:batch]), (name, batch)
            assert np.array_equal(gpu.last_query_stats(batch), cst[:batch].astype(np.uint32)), (name, batch)
        one = gpu.search(Q[3], 10, 72)
        assert np.array_equal(one, ck[3][:len(one)]), name


def test_search_matches_reference_order_within_tolerance():
    """Against the reference-order metric (what usearch computes): identical ids on tie-free data, distances within
    1e-5 relative."""
    n, dim = 3000, 96
    X, Q = gc.make_data(n, dim, "l2sq", 77)
    ref_order = CpuIndex(load_oracle(), dim, "l2sq", order=0, wave=0)
    ref_order.reserve(n)
    ref_order.add_many(np.arange(n), X)
    gpu = gc.gpu_index(dim, "l2sq")
    gpu.load(ref_order.save())
    gk, gd, _ = gpu.search_batch(Q, 10)
    ck, cd, _, _ = ref_order.search_many(Q, 10)
    assert np.all(np.abs(gd - cd) <= 1e-5 * np.abs(cd))
    assert np.mean(gk == ck) > 0.99  # summation-order near-ties may swap neighbours of (almost) equal distance
    for i in range(len(Q)):
        if not np.array_equal(gk[i], ck[i]):
            assert sorted(gk[i]) == sorted(ck[i]) or np.min(np.abs(np.diff(cd[i]))) <= 1e-5 * np.max(cd[i])


# ------------------------------------------------------------------------------------------------- build
BUILD_CASES = [(400, 8, "l2sq", 4, 8, 24, 1, 1), (1500, 16, "l2sq", 16, 32, 128, 1, 1),
               (3000, 16, "l2sq", 16, 32, 128, 256, 8), (3000, 24, "cosine", 8, 16, 64, 512, 4),
               (2500, 40, "ip", 16, 32, 100, 128, 16), (1200, 768, "l2sq", 16, 32, 128, 256, 8),
               (2000, 128, "cosine", 16, 32, 128, 1024, 2),
               # 2 and 4 chunks per lane (dimensions 512 / 1024: unrolled instantiations since round 3)
               (1000, 512, "cosine", 16, 32, 96, 256, 8), (900, 1024, "ip", 12, 24, 64, 128, 8),
               # ef_construction below the list capacities: the insert search is bounded by ef_construction itself
               (1000, 16, "l2sq", 16, 32, 8, 256, 8), (600, 12, "cosine", 8, 16, 6, 1, 1)]


@pytest.mark.parametrize("n,dim,metric,M,M0,efc,max_batch,growth_div", BUILD_CASES)
def test_bulk_build_graph_is_bit_identical_to_oracle(n, dim, metric, M, M0, efc, max_batch, growth_div):
    """The GPU batch-synchronous build against its CPU restatement: the serialized graph (levels, every neighbour
    list in order, keys, vectors) must be byte-identical.  With max_batch = 1 the restatement IS the reference's
    sequential add() (tests/test_oracle_golden.py), so those cases pin the kernels to the reference algorithm."""
    X, Q = gc.make_data(n, dim, metric, 300 + n + dim)
    keys = np.arange(n, dtype=np.int64) * 7 + 5
    cpu = gc.oracle_index(dim, metric, M, M0, efc)
    cpu.reserve(n)
    cpu.build_batch(keys, X, max_batch, growth_div)
    gpu = gc.gpu_index(dim, metric, M, M0, efc)
    gpu.reserve(n)
    gpu.set_build_params(max_batch, growth_div)
    for c in range(0, n, 2048):  # DuckDB hands chunks of <= 2048 rows
        gpu.stage(keys[c:c + 2048], X[c:c + 2048])
    gpu.build_finalize()
    diff = gc.first_graph_difference(gpu.save(), cpu.save())
    assert diff is None, diff
    assert (gpu.size(), gpu.capacity(), gpu.max_level()) == (cpu.size(), cpu.capacity(), cpu.max_level())
    gk, gd, _ = gpu.search_batch(Q, 10)
    ck, cd, _, _ = cpu.search_many(Q, 10)
    assert np.array_equal(gk, ck) and np.array_equal(_bits(gd), _bits(cd))


def test_incremental_add_after_bulk_build():
    """HNSWIndex::Construct path (hnsw_index.cpp:421-479): chunks appended after a bulk load, with power-of-two
    reserve growth restarting the level generator exactly like usearch does."""
    n0, n1, dim = 1000, 600, 12
    X, Q = gc.make_data(n0 + n1, dim, "l2sq", 909)
    cpu, gpu = gc.oracle_index(dim, "l2sq", 8, 16, 40), gc.gpu_index(dim, "l2sq", 8, 16, 40)
    for ix in (cpu, gpu):
        ix.reserve(n0)
    gpu.set_build_params(128, 8)
    cpu.build_batch(np.arange(n0), X[:n0], 128, 8)
    gpu.add(np.arange(n0), X[:n0])
    cap = n0
    for c in range(n0, n0 + n1, 200):
        if c + 200 > cap:
            cap = 1 << int(np.ceil(np.log2(c + 200)))
            cpu.reserve(cap), gpu.reserve(cap)
        cpu.build_batch(np.arange(c, c + 200), X[c:c + 200], 128, 8)
        gpu.add(np.arange(c, c + 200), X[c:c + 200])
    diff = gc.first_graph_difference(gpu.save(), cpu.save())
    assert diff is None, diff


def test_readme_example():
    """README / hnsw_result.test: 9^3 grid, l2sq, query [1,2,3], k=3 -> distances 0, 1, 1 (array_distance 0,1,1)."""
    X = datagen.readme_grid()
    gpu = gc.gpu_index(3, "l2sq")
    gpu.reserve(len(X))
    gpu.add(np.arange(len(X)), X)
    keys, d, cnt = gpu.search_batch(np.array([[1, 2, 3]], dtype=np.float32), 3)
    assert cnt[0] == 3 and d[0].tolist() == [0.0, 1.0, 1.0]
    assert np.array_equal(X[keys[0][0]], [1, 2, 3])
    got = gc.pkg().distance_batch("array_distance", X[keys[0]], np.array([1, 2, 3], dtype=np.float32))
    assert got.tolist() == [0.0, 1.0, 1.0]


def test_null_rows_capacity_and_empty_inputs():
    dim = 8
    X, Q = gc.make_data(300, dim, "l2sq", 5)
    gpu = gc.gpu_index(dim, "l2sq")
    k, d, cnt = gpu.search_batch(Q[:4], 5)  # empty index: no results
    assert cnt.tolist() == [0, 0, 0, 0] and np.all(k == -1)
    gpu.reserve(200)
    validity = np.full((300 + 63) // 64, np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    null_rows = [0, 5, 64, 65, 199]
    for r in null_rows:
        validity[r // 64] &= ~np.uint64(1 << (r % 64))
    gpu.add(np.arange(200), X[:200], validity[:4])  # NULL vectors are skipped (hnsw_index.cpp:467-470)
    assert gpu.size() == 200 - len(null_rows)
    found = gpu.search_batch(X[[0, 5, 64, 1, 2]], 1)[0][:, 0]
    assert found[3] == 1 and found[4] == 2 and not set(found[:3].tolist()) & set(null_rows)
    with pytest.raises(gc.pkg().VssError, match="Reserve capacity ahead of insertions!"):
        gpu.add(np.arange(200, 300), X[200:300])
    gpu.add(np.arange(0), X[:0])  # empty chunk is a no-op
    assert gpu.size() == 195


# ------------------------------------------------------------------------------------------------- exact search
@pytest.mark.parametrize("n,dim,metric", [(5000, 64, "l2sq"), (3000, 768, "cosine"), (4000, 100, "ip"), (729, 3, "l2sq")])
def test_exact_search(n, dim, metric):
    """exact=true (usearch search_exact_): ids equal the oracle's brute force wherever distances are distinct, and
    distances carry the wave-order bits."""
    if dim == 3:
        X, Q = datagen.readme_grid(), np.array([[1.2, 2.1, 3.3], [8.7, 1.1, 4.9]], dtype=np.float32)
    else:
        X, Q = gc.make_data(n, dim, metric, 40 + dim, nq=33)
    cpu, gpu = gc.oracle_index(dim, metric), gc.gpu_index(dim, metric)
    cpu.reserve(len(X)), gpu.reserve(len(X))
    cpu.build_batch(np.arange(len(X)), X, 256, 8)
    gpu.set_build_params(256, 8)
    gpu.add(np.arange(len(X)), X)
    for k in (1, 10, 50):
        gk, gd, gcnt = gpu.search_batch(Q, k, exact=True)
        ck, cd, ccnt, _ = cpu.search_many(Q, k, exact=True)
        assert np.array_equal(gcnt, ccnt)
        assert np.array_equal(_bits(gd), _bits(cd))
        for i in range(len(Q)):
            distinct = len(set(cd[i].tolist())) == k and (k == len(cd[i]))
            if distinct:
                assert np.array_equal(gk[i], ck[i])


@pytest.mark.parametrize("metric,dim,nq", [("l2sq", 24, 96), ("cosine", 24, 96), ("l2sq", 64, 200), ("ip", 32, 130), ("cosine", 128, 257)])
def test_exact_search_over_many_chunks_with_the_select_folded_into_the_score_tile(metric, dim, nq, monkeypatch):
    """Round 4: from the second 32768-row chunk on the score tile's epilogue keeps only the scores that beat a query's K'-th
    best so far, and the running top-K' is refreshed from those survivors every eight chunks.  The answers (ids, distance
    bits, counts) must be those of the plain path (VSS_EXACT_FILTER=0: every score stored, a select after every chunk) and
    of brute force in float64 — over 9+ chunks with deletions; and when the rows arrive in DESCENDING distance (every row of
    every chunk beats the threshold: the survivor buffers overflow) the search is redone the plain way, same answers.
    Round 5: dimensions that are a multiple of 32 take the persistent tile with LDS-DMA operands (k_exact_scores_v4), the others
    the register-staged persistent tile (v3); query counts that leave a ragged last query tile beside whole ones, a table whose
    last row tile is ragged; and round 3's one-tile kernel (VSS_EXACT_KERNEL=2) must give the same bits as either."""
    n = 300_000
    rng = np.random.default_rng(77)
    X = rng.standard_normal((n, dim)).astype(np.float32)
    Q = rng.standard_normal((nq, dim)).astype(np.float32)
    if metric != "l2sq":
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    dead = rng.choice(n, 5000, replace=False)

    def answers(rows, filt, kernel=None):
        monkeypatch.setenv("VSS_EXACT_FILTER", "1" if filt else "0")
        if kernel:
            monkeypatch.setenv("VSS_EXACT_KERNEL", str(kernel))
        else:
            monkeypatch.delenv("VSS_EXACT_KERNEL", raising=False)
        gpu = gc.gpu_index(dim, metric, 8, 16, 16)  # (a cheap graph: only the exact path is under test)
        gpu.reserve(n)
        gpu.set_build_params(32768, 4)
        gpu.add(np.arange(n), rows)
        assert gpu.remove(dead) == len(dead)
        # k = 200: K' = 208, windows sized from the rows already seen (2.4, then 8 chunks); k = 600: beyond CAND_CAP / 8 survivors
        # per chunk even in the best case, the plain way from the start (round 5)
        out = [gpu.search_batch(Q, k, exact=True) for k in (1, 10, 100, 200, 600)]
        gpu.close()
        return out

    plain, folded = answers(X, False), answers(X, True)
    for other in (folded, answers(X, True, kernel=2), answers(X, False, kernel=4)):
        for (pk, pd, pc), (fk, fd, fc) in zip(plain, other):
            assert np.array_equal(pk, fk) and np.array_equal(_bits(pd), _bits(fd)) and np.array_equal(pc, fc)
    monkeypatch.delenv("VSS_EXACT_KERNEL", raising=False)
    # against brute force in float64 (ids wherever the float64 distances are not within float32 noise of each other)
    live = np.ones(n, dtype=bool)
    live[dead] = False
    Xd, Qd = X.astype(np.float64), Q.astype(np.float64)
    best = np.full((nq, 100), np.inf)
    best_i = np.full((nq, 100), -1, dtype=np.int64)
    for lo in range(0, n, 50_000):
        blk = Xd[lo:lo + 50_000]
        dd = (Qd ** 2).sum(1)[:, None] - 2 * Qd @ blk.T + (blk ** 2).sum(1)[None, :] if metric == "l2sq" else 1.0 - Qd @ blk.T
        dd[:, ~live[lo:lo + 50_000]] = np.inf
        alld = np.concatenate([best, dd], axis=1)
        alli = np.concatenate([best_i, np.broadcast_to(np.arange(lo, lo + len(blk)), dd.shape)], axis=1)
        order = np.argsort(alld, axis=1, kind="stable")[:, :100]
        best, best_i = np.take_along_axis(alld, order, 1), np.take_along_axis(alli, order, 1)
    fk, fd, fc = folded[2]
    assert np.all(fc == 100) and not np.isin(fk, dead).any()
    agree = np.mean([len(set(fk[i].tolist()) & set(best_i[i].tolist())) / 100 for i in range(nq)])
    assert agree >= 0.999, agree  # (float32 scores against float64 ranking: only near-ties at the 100th place may differ)
    assert np.allclose(fd, best, rtol=2e-5, atol=2e-6)
    # adversarial order: every chunk is closer to the queries than everything before it -> overflow -> redone the plain way
    q0 = Q[:1]
    order = np.argsort(-((X - q0) ** 2).sum(1)) if metric == "l2sq" else np.argsort(X @ q0[0])
    Xs = np.ascontiguousarray(X[order])
    Qs = np.repeat(q0, 8, axis=0)

    def one(rows, filt):
        monkeypatch.setenv("VSS_EXACT_FILTER", "1" if filt else "0")
        gpu = gc.gpu_index(dim, metric, 8, 16, 16)
        gpu.reserve(n)
        gpu.set_build_params(32768, 4)
        gpu.add(np.arange(n), rows)
        out = gpu.search_batch(Qs, 10, exact=True)
        gpu.close()
        return out
    (ak, ad, ac), (bk, bd, bc) = one(Xs, False), one(Xs, True)
    assert np.array_equal(ak, bk) and np.array_equal(_bits(ad), _bits(bd)) and np.array_equal(ac, bc)
    assert set(ak[0].tolist()) == set(range(n - 10, n))  # the ten nearest rows are the last ten of the sorted table


# ------------------------------------------------------------------------------------------------- deletes / compact / stream
def test_tombstones_search_compact_and_streams():
    n, dim = 2500, 32
    X, Q = gc.make_data(n, dim, "l2sq", 1234)
    cpu, gpu = gc.oracle_index(dim, "l2sq"), gc.gpu_index(dim, "l2sq")
    cpu.reserve(n), gpu.reserve(n)
    cpu.build_batch(np.arange(n), X, 256, 8)
    gpu.set_build_params(256, 8)
    gpu.add(np.arange(n), X)
    dead = np.arange(0, n, 3)
    assert gpu.remove(np.concatenate([dead, [10 ** 9]])) == len(dead)
    assert gpu.remove(dead[:5]) == 0
    for k in dead:
        cpu.remove(int(k))
    # the engine reports the true live count; usearch's free ring mis-reports it once 64 slots were freed
    # (ring_gt::size() returns 0 when full, index.hpp:1202-1209 — the oracle reproduces that)
    assert gpu.size() == n - len(dead) and gpu.nodes() == cpu.nodes() == n
    gk, gd, gcnt = gpu.search_batch(Q, 10, 40)
    ck, cd, ccnt, _ = cpu.search_many(Q, 10, ef=40)
    assert np.array_equal(gk, ck) and np.array_equal(_bits(gd), _bits(cd)) and np.array_equal(gcnt, ccnt)
    assert not set(gk.ravel().tolist()) & set(dead.tolist())
    # the stream written by the GPU engine is the reference's format: the CPU restatement (and usearch) load it
    blob = gpu.save()
    cblob = cpu.save()
    head = 8 + n * dim * 4  # count_present / count_deleted: the reference mis-counts once its free ring wrapped
    assert blob[:head + 17] == cblob[:head + 17] and blob[head + 33:] == cblob[head + 33:]
    assert np.frombuffer(blob[head + 17:head + 33], dtype=np.uint64).tolist() == [n - len(dead), len(dead)]
    back = gc.oracle_index(dim, "l2sq")
    back.load(blob)
    assert np.array_equal(back.search_many(Q, 10, ef=40)[0], gk)
    # compact drops the tombstones (documented behaviour, README.md:69) and keeps answers of live rows reachable
    gpu.compact()
    assert gpu.size() == gpu.nodes() == n - len(dead)
    ek, _, _ = gpu.search_batch(Q, 10, exact=True)
    ak, ad, _ = gpu.search_batch(Q, 10, 128)
    assert gc.recall_at_k(ak, ek) > 0.9
    assert not set(ak.ravel().tolist()) & set(dead.tolist())
    g = parse_stream(gpu.save())
    assert g["rows"] == n - len(dead) and np.all(g["keys"] != np.iinfo(np.int64).max)


@pytest.mark.parametrize("max_batch,growth_div", [(1, 1), (64, 4)])
def test_removed_slots_are_reused_through_the_update_path(max_batch, growth_div):
    """usearch hands tombstoned slots back to later inserts (free ring, index_dense.hpp:1766-1771) and re-links them
    with update() (index.hpp:2801-2859): lists blanked, searched from the global entry while the OLD vector is still in
    the slot, key and vector replaced afterwards, no level drawn.  With max_batch = 1 the oracle's build_batch is the
    reference's add() sequence (tests/test_oracle_golden.py::test_batch_build_with_reuse_*), so the (1, 1) case pins
    which slot every row lands in and every list, byte for byte; (64, 4) pins the batched variant of the same path."""
    n0, dim = 1200, 16
    X, Q = gc.make_data(2400, dim, "l2sq", 777)
    cpu, gpu = gc.oracle_index(dim, "l2sq", 8, 16, 40), gc.gpu_index(dim, "l2sq", 8, 16, 40)
    cpu.reserve(4096), gpu.reserve(4096)
    gpu.set_build_params(max_batch, growth_div)

    def add(keys, vecs):
        cpu.build_batch(keys, vecs, max_batch, growth_div)
        gpu.add(keys, vecs)

    def remove(keys):
        assert gpu.remove(np.asarray(keys, dtype=np.int64)) == len(keys)
        for k in keys:
            assert cpu.remove(int(k)) == 1

    def check(what):
        diff = gc.first_graph_difference(gpu.save(), cpu.save(), ignore_counts=True)
        assert diff is None, "%s: %s" % (what, diff)
        gk, gd, gcnt = gpu.search_batch(Q, 10, 40)
        ck, cd, ccnt, _ = cpu.search_many(Q, 10, ef=40)
        assert np.array_equal(gk, ck) and np.array_equal(_bits(gd), _bits(cd)) and np.array_equal(gcnt, ccnt), what

    add(np.arange(n0), X[:n0])
    # 50 deletions (the ring has not wrapped) including the entry node, then 80 inserts: 50 take over slots, 30 append
    entry_key = cpu.node_key(cpu.entry_slot())
    dead = sorted(set([entry_key] + list(range(5, 250, 5))))
    remove(dead)
    add(10_000 + np.arange(80), X[n0:n0 + 80])
    assert gpu.nodes() == cpu.nodes() == n0 + 80 - len(dead)
    assert gpu.size() == gpu.nodes()
    check("after the first reuse round")
    # 200 deletions wrap the 64-entry ring (usearch quirk Q11: only some of the slots come back), inserts in DuckDB chunks
    dead2 = list(range(300, 1100, 4))
    remove(dead2)
    for c in range(0, 260, 100):
        m = min(100, 260 - c)
        add(20_000 + c + np.arange(m), X[n0 + 80 + c:n0 + 80 + c + m])
    assert gpu.nodes() == cpu.nodes()
    check("after the ring wrapped")
    # duplicate keys are refused once the key map exists (it does after a remove), and the refused call consumes
    # nothing from the ring: the next rounds still agree slot for slot
    remove(list(range(1101, 1161, 2)))
    with pytest.raises(RuntimeError):
        gpu.add(np.array([20_001]), X[:1])
    add(np.array([40_000, 40_001]), X[2100:2102])
    check("after a refused duplicate")
    # a loaded index rebuilds its free list from the stream's tombstones, in slot order (index_dense.hpp:1901-1930)
    blob = gpu.save()
    assert np.count_nonzero(parse_stream(blob)["keys"] == np.iinfo(np.int64).max) >= 28
    cpu, gpu = gc.oracle_index(dim, "l2sq", 8, 16, 40), gc.gpu_index(dim, "l2sq", 8, 16, 40)
    cpu.load(blob), gpu.load(blob)
    cpu.reserve(4096), gpu.reserve(4096)
    gpu.set_build_params(max_batch, growth_div)
    add(30_000 + np.arange(60), X[2000:2060])
    check("after load + reuse")


def test_filtered_search_pushes_the_predicate_into_the_traversal():
    """usearch filtered_search semantics (bit-exact vs the oracle, which is bit-exact vs the reference): only admitted
    rows are returned, rejected rows are still traversed, tombstones stay excluded; k admitted rows come back even when
    the predicate is rare (the reference's filter-above-the-scan plan returns fewer: SURVEY quirk Q9)."""
    import golden_cases
    n, dim = 4000, 32
    X, Q = gc.make_data(n, dim, "l2sq", 515, nq=48)
    cpu, gpu = gc.oracle_index(dim, "l2sq"), gc.gpu_index(dim, "l2sq")
    cpu.reserve(n), gpu.reserve(n)
    keys = np.arange(n, dtype=np.int64) * 2 + 1
    cpu.build_batch(keys, X, 256, 8)
    gpu.set_build_params(256, 8)
    gpu.add(keys, X)
    dead = keys[::9]
    gpu.remove(dead)
    for key in dead:
        cpu.remove(int(key))
    n_bits = 2 * n - 3
    for frac, k, ef in ((0.5, 10, 64), (0.03, 10, 48), (0.9, 20, 200)):
        bm = golden_cases.filter_bitmap(n_bits, 31 + k, frac)
        gk, gd, gcnt = gpu.search_batch_filtered(Q, k, ef, bm, n_bits)
        ck, cd, ccnt, _ = cpu.search_many_filtered(Q, k, ef, bm, n_bits)
        assert np.array_equal(gk, ck) and np.array_equal(_bits(gd), _bits(cd)) and np.array_equal(gcnt, ccnt)
        live = gk[gk >= 0]
        assert np.all((bm[live >> 6] >> (live & 63).astype(np.uint64)) & np.uint64(1) == 1)
        assert not set(live.tolist()) & set(dead.tolist())
    assert np.all(gcnt == 20)


def _merge_reference(d, ids, k):
    """ascending (distance, row id) over the valid cells of the union; -1 / +inf beyond (vssgpu.h: vss_merge_topk_device)"""
    G, B, _ = d.shape
    flat_d = np.transpose(d, (1, 0, 2)).reshape(B, -1)
    flat_i = np.transpose(ids, (1, 0, 2)).reshape(B, -1)
    out_d = np.full((B, k), np.inf, dtype=np.float32)
    out_i = np.full((B, k), -1, dtype=np.int64)
    cnt = np.zeros(B, dtype=np.uint32)
    for q in range(B):
        ok = flat_i[q] >= 0
        dd, ii = flat_d[q][ok], flat_i[q][ok]
        order = np.lexsort((ii, dd))[:k]
        out_d[q, :len(order)] = dd[order]
        out_i[q, :len(order)] = ii[order]
        cnt[q] = len(order)
    return out_d, out_i, cnt


@pytest.mark.parametrize("G,B,k", [(4, 37, 10), (8, 33, 100), (8, 5, 2047), (3, 70, 1), (1, 9, 10), (8, 20, 64)])
@pytest.mark.parametrize("flavour", ["plain", "ties", "unsorted"])
def test_merge_topk_kernel(G, B, k, flavour):
    """k_merge_topk (round 6: co-ranking over ascending per-shard lists staged in LDS; 8 x 2047 exceeds the LDS budget and probes
    the global arrays): shards with short lists (padded cells), an empty shard, exact distance ties inside and across shards
    (the row id decides), and lists that break the ascending promise (the kernel checks it and counts instead)."""
    lib = gc.pkg().load_library()  # imports torch first (one HIP runtime per process)
    import torch
    rng = np.random.default_rng(3 + G * 1000 + k)
    if flavour == "ties":  # a coarse lattice: many equal distances, inside a shard and across shards
        d = np.sort((rng.integers(0, max(2, k // 3 + 2), (G, B, k)) / 8.0).astype(np.float32), axis=2)
    else:
        d = np.sort(rng.random((G, B, k)).astype(np.float32), axis=2)
    ids = rng.permutation(G * B * k).reshape(G, B, k).astype(np.int64)
    if G > 1:  # shard 1 answers with fewer rows (a small shard / a selective predicate); the last shard is empty for some queries
        short = (k * 7) // 10
        d[1, :, short:] = np.inf
        ids[1, :, short:] = -1
        d[G - 1, ::3, :] = np.inf
        ids[G - 1, ::3, :] = -1
    if flavour == "unsorted":  # not what the contract promises: shuffled cells, unused cells in the middle
        for g in range(G):
            for q in range(0, B, 2):
                perm = rng.permutation(k)
                d[g, q], ids[g, q] = d[g, q][perm], ids[g, q][perm]
    td, ti = torch.from_numpy(d).cuda(), torch.from_numpy(ids).cuda()
    od, oi = torch.empty((B, k), dtype=torch.float32, device="cuda"), torch.empty((B, k), dtype=torch.int64, device="cuda")
    oc = torch.empty(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc = lib.vss_merge_topk_device(td.data_ptr(), ti.data_ptr(), G, B, k, od.data_ptr(), oi.data_ptr(), oc.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    rd, ri, rcnt = _merge_reference(d, ids, k)
    assert np.array_equal(oi.cpu().numpy(), ri)
    assert np.array_equal(_bits(od.cpu().numpy()), _bits(rd))
    assert np.array_equal(oc.cpu().numpy().astype(np.uint32), rcnt)
    # the packed layout of one all-gather per launch: same cells, block strides
    block = gc.pkg().load_library().vss_packed_block_bytes(B, k)
    packed = np.zeros((G, block), dtype=np.uint8)
    for g in range(G):
        packed[g, :B * k * 8] = ids[g].reshape(-1).view(np.uint8)
        packed[g, B * k * 8:B * k * 12] = d[g].reshape(-1).view(np.uint8)
    tp = torch.from_numpy(packed).cuda()
    od.fill_(0), oi.fill_(0), oc.fill_(0)
    torch.cuda.synchronize()
    assert lib.vss_merge_topk_packed_device(tp.data_ptr(), G, B, k, od.data_ptr(), oi.data_ptr(), oc.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(oi.cpu().numpy(), ri) and np.array_equal(_bits(od.cpu().numpy()), _bits(rd))
    assert np.array_equal(oc.cpu().numpy().astype(np.uint32), rcnt)


def test_wide_lists_large_ef_and_hbm_visited_set():
    """Paths the default options never take: neighbour lists wider than one wave (M0 = 80), ef above 256 (8-register
    candidate list, visited set in HBM), k above 64, tombstones combined with a large ef."""
    n, dim = 2500, 20
    X, Q = gc.make_data(n, dim, "l2sq", 4321, nq=40)
    cpu, gpu = gc.oracle_index(dim, "l2sq", 40, 80, 200), gc.gpu_index(dim, "l2sq", 40, 80, 200)
    cpu.reserve(n), gpu.reserve(n)
    cpu.build_batch(np.arange(n), X, 300, 6)
    gpu.set_build_params(300, 6)
    gpu.add(np.arange(n), X)
    diff = gc.first_graph_difference(gpu.save(), cpu.save())
    assert diff is None, diff
    for k, ef in ((10, 300), (100, 500), (70, 0), (10, 130)):
        gk, gd, gcnt = gpu.search_batch(Q, k, ef)
        ck, cd, ccnt, cst = cpu.search_many(Q, k, ef=ef if ef else None)
        assert np.array_equal(gk, ck) and np.array_equal(_bits(gd), _bits(cd)) and np.array_equal(gcnt, ccnt)
        assert np.array_equal(gpu.last_query_stats(len(Q)), cst.astype(np.uint32))
    dead = np.arange(1, n, 2)
    gpu.remove(dead)
    for key in dead:
        cpu.remove(int(key))
    for k, ef in ((10, 300), (50, 64)):
        gk, gd, gcnt = gpu.search_batch(Q, k, ef)
        ck, cd, ccnt, _ = cpu.search_many(Q, k, ef=ef)
        assert np.array_equal(gk, ck) and np.array_equal(_bits(gd), _bits(cd)) and np.array_equal(gcnt, ccnt)
    for k, ef in ((10, 600), (600, 0), (2000, 1024)):  # beyond the register lists: the list lives in HBM
        gk, gd, gcnt = gpu.search_batch(Q, k, ef)
        ck, cd, ccnt, _ = cpu.search_many(Q, k, ef=ef if ef else None)
        assert np.array_equal(gk, ck) and np.array_equal(_bits(gd), _bits(cd)) and np.array_equal(gcnt, ccnt)


def test_pipelined_contexts_equal_blocking_calls():
    """vss_search_batch_device_begin/_end on several contexts return exactly what the blocking call returns."""
    lib = gc.pkg().load_library()
    import torch
    n, dim, B, k = 20000, 64, 256, 10
    X, Q = gc.make_data(n, dim, "cosine", 77, nq=3 * B)
    gpu = gc.gpu_index(dim, "cosine")
    gpu.reserve(n)
    gpu.add(np.arange(n), X)
    ref = [gpu.search_batch(Q[i * B:(i + 1) * B], k, 96) for i in range(3)]
    dq = torch.from_numpy(Q).cuda()
    outs = [(torch.empty((B, k), dtype=torch.int64, device="cuda"), torch.empty((B, k), dtype=torch.float32, device="cuda"),
             torch.empty(B, dtype=torch.int32, device="cuda")) for _ in range(3)]
    torch.cuda.synchronize()
    for c in range(3):
        a, b, cc = outs[c]
        gpu.search_begin(c, dq[c * B:(c + 1) * B].data_ptr(), B, k, 96, a.data_ptr(), b.data_ptr(), cc.data_ptr())
    with pytest.raises(gc.pkg().VssError, match="already has a batch in flight"):
        gpu.search_begin(1, dq.data_ptr(), B, k, 96, outs[1][0].data_ptr(), outs[1][1].data_ptr(), outs[1][2].data_ptr())
    for c in (2, 0, 1):
        gpu.search_end(c)
    torch.cuda.synchronize()
    for c in range(3):
        assert np.array_equal(outs[c][0].cpu().numpy(), ref[c][0])
        assert np.array_equal(_bits(outs[c][1].cpu().numpy()), _bits(ref[c][1]))


def test_two_rank_sharded_bench():
    """The N > 1 path of bench.py as real processes: two ranks (both on this box's single GPU, gloo instead of RCCL)
    build their row-range shards, run the pipelined probe with all-gather + merge kernel, and report merged recall."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VSS_BENCH_SAME_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(root, "bench.py"), "--gpus", "2", "--rows", "300000", "--dim", "128",
           "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["config"]["parallelism"] == "shard2"
    assert res["recall_at_10"] >= 0.95 and res["value"] > 0


# ------------------------------------------------------------------------------------------------- size-independent properties
def test_one_rank_rccl_collective_through_bench():
    """The exchange of the N > 1 path on RCCL itself, as far as one GPU allows: a one-rank `nccl` process group (RCCL init),
    the packed all_gather_into_tensor of every launch issued on the side stream, the packed merge kernel — the code path the
    driver's 2/4/8-GPU runs take, minus the peers (the first RCCL init must not happen in the scaling run)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VSS_BENCH_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--rows", "200000", "--dim", "64", "--steps", "6",
           "--warmup", "3", "--coalesce", "2", "--no-cpu-baseline", "--extras", "none", "--heldout-batches", "1",
           "--host-api-seconds", "0", "--regimes", "none", "--no-small-launches"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["rccl_ranks"] == 1 and res["collective_backend"] == "nccl"
    assert res["collectives_per_launch"] == 1 and res["collectives_timed"] == 3  # 6 steps = 3 launches of 2 batches
    assert res["config"]["parallelism"] == "shard1" and res["recall_at_10"] >= 0.95 and res["value"] > 0


def test_one_rank_rccl_replicated_mode_through_bench():
    """Round 5: the OTHER multi-GPU mode of bench.py — one full index per GPU, every rank its own query
    batches, no data-path collective — as far as one GPU allows: a one-rank `nccl` process group, the run's reductions (slowest
    rank's time, worst recall, the common ef_search, the device roll call) over RCCL, the compact last line with the mode's keys."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VSS_BENCH_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29643", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--mode", "replicated", "--rows", "200000", "--dim", "64",
           "--steps", "6", "--warmup", "3", "--coalesce", "2", "--no-cpu-baseline", "--extras", "none", "--heldout-batches", "1",
           "--host-api-seconds", "0", "--regimes", "none", "--no-small-launches"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines[-1]) < 4000  # the compact last line (tests/test_bench_helpers.py holds the rule)
    res = json.loads(lines[-1])
    assert res["rccl_ranks"] == 1 and res["collective_backend"] == "nccl" and res["multi_gpu_mode"] == "replicated"
    assert res["collectives_per_launch"] == 0 and res["collectives_timed"] == 0  # no exchange on the data path
    assert res["config"]["parallelism"] == "replica1" and res["scaling"] == "weak" and len(res["rank_pci"]) == 1
    assert res["recall_at_10"] >= 0.95 and res["value"] > 0 and res["roofline"]["frac"] > 0


def test_properties_at_scale():
    """BASELINE-shaped data at a size the oracle could not finish in seconds: structural invariants of the graph,
    sortedness, idempotence, recall against the exact path."""
    n, dim, nq = 200_000, 128, 512
    X = datagen.mixture(n, dim, 31337, intrinsic_dim=16, basis_seed=31337, centre_scale=0.1)
    Q = datagen.mixture(nq, dim, 31338, n_clusters=int(np.sqrt(n)), intrinsic_dim=16, basis_seed=31337, centre_scale=0.1)
    gpu = gc.gpu_index(dim, "l2sq")
    gpu.reserve(n)
    for c in range(0, n, 50_000):
        gpu.stage(np.arange(c, min(n, c + 50_000)), X[c:c + 50_000])
    gpu.build_finalize()
    assert gpu.size() == n
    k1, d1, c1 = gpu.search_batch(Q, 10, 64)
    k2, d2, c2 = gpu.search_batch(Q, 10, 64)
    assert np.array_equal(k1, k2) and np.array_equal(_bits(d1), _bits(d2))        # idempotent
    assert np.all(np.diff(d1, axis=1) >= 0) and np.all(c1 == 10)                    # ascending
    for i in range(0, nq, 37):                                                       # distances are the real ones
        ref = ((X[k1[i]].astype(np.float64) - Q[i]) ** 2).sum(1)
        assert np.all(np.abs(d1[i] - ref) <= 1e-5 * ref)
    ek, ed, _ = gpu.search_batch(Q, 10, exact=True)
    assert np.all(ed[:, 0] <= d1[:, 0] * (1 + 1e-6))
    assert gc.recall_at_k(k1, ek) > 0.9
    g = parse_stream(gpu.save())
    deg0 = np.array([len(a[0]) for a in g["adj"]])
    assert deg0.max() <= 32 and deg0.min() >= 1
    for s in range(0, n, 997):
        for lvl, nb in enumerate(g["adj"][s]):
            assert len(set(nb.tolist())) == len(nb) and s not in nb
            assert np.all(g["levels"][nb] >= lvl)
    lv = np.zeros(n, dtype=np.int16)
    load_oracle().orc_draw_levels(16, n, lv.ctypes.data)
    assert np.array_equal(g["levels"], lv)                                           # the reference's level sequence


def test_properties_at_full_benchmark_size():
    """BASELINE configs[2] at its full size — 10M rows x FLOAT[768], cosine, top-10, 1024-query batches, the options of
    the bench run — checked through properties that need no CPU replay: the level histogram is the reference generator's,
    every list is within its capacity, the search is idempotent, complete and ascending, every distance it reports for a
    row the exact path also returns carries the same bits, and recall@10 against the exact path is the benchmark's: the
    operating point is chosen by bench.py's own rule on two selection batches (bench.select_ef over bench.EF_SWEEP), the
    properties are checked at THAT ef_search, and the recall bar (>= 0.95) is held on eight batches the selection never saw."""
    import torch
    sys.path.insert(0, gc.ROOT)
    import bench
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 60 << 30:
        pytest.skip("needs 60 GB of free device memory")
    n, dim, B, k = 10_000_000, 768, 1024, 10
    M, efc = bench.HEADLINE_OPTIONS["M"], bench.HEADLINE_OPTIONS["ef_construction"]
    gen = bench.Mixture(n, dim, True, dev)
    gpu = gc.pkg().GpuIndex(dim, "cosine", M, 2 * M, efc)
    gpu.reserve(n)
    for c in range(0, n, bench.CHUNK):
        m = min(bench.CHUNK, n - c)
        x = gen.rows(bench.DATA_SEED, c // bench.CHUNK, m)
        ids = torch.arange(c, c + m, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        gpu.stage_device(ids.data_ptr(), x.data_ptr(), m)
        del x, ids
    gpu.build_finalize()
    assert gpu.size() == gpu.nodes() == n
    # levels: the reference's generator, draw for draw (compared as a histogram: the per-level node counts)
    lv = np.zeros(n, dtype=np.int16)
    load_oracle().orc_draw_levels(M, n, lv.ctypes.data)
    assert gpu.max_level() == int(lv.max())
    for level in range(int(lv.max()) + 1):
        nodes, edges, _, _ = (int(v) for v in gpu.level_stats(level))
        assert nodes == int(np.count_nonzero(lv >= level)), level
        assert nodes - 1 <= edges <= nodes * (2 * M if level == 0 else M), level    # connected, within capacity
    out = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
            torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(3)]

    def answer(qq, e, exact=False):
        gpu.search_batch_device(qq.data_ptr(), B, k, e, out[0][0].data_ptr(), out[0][1].data_ptr(), out[0][2].data_ptr(), exact=exact)
        torch.cuda.synchronize()
        return out[0][0].clone()

    # the bench's operating point: selection on batches 0-1, the bar on eight held-out batches
    sel = [gen.rows(bench.QUERY_SEED, i, B) for i in range(2)]
    torch.cuda.synchronize()
    sel_truth = [answer(qq, 0, True) for qq in sel]
    ef, sel_mean, sel_se, _ = bench.select_ef(
        lambda e: sum((bench.recall_per_query(answer(qq, e), t) for qq, t in zip(sel, sel_truth)), []), bench.EF_SWEEP, 0.95)
    assert sel_mean - 2 * sel_se >= 0.95, (ef, sel_mean, sel_se)
    held = []
    for i in range(8):
        qq = gen.rows(bench.QUERY_SEED, 5000 + i, B)
        torch.cuda.synchronize()
        held += bench.recall_per_query(answer(qq, ef), answer(qq, 0, True))
    held_mean, held_se = bench.mean_and_se(held)
    assert held_mean >= 0.95, (ef, held_mean, held_se)
    q = sel[0]
    for i in (0, 1):
        gpu.search_batch_device(q.data_ptr(), B, k, ef, out[i][0].data_ptr(), out[i][1].data_ptr(), out[i][2].data_ptr())
    gpu.search_batch_device(q.data_ptr(), B, k, 0, out[2][0].data_ptr(), out[2][1].data_ptr(), out[2][2].data_ptr(),
                            exact=True)
    torch.cuda.synchronize()
    (k1, d1, c1), (k2, d2, _), (ek, ed, ec) = [tuple(t.cpu().numpy() for t in o) for o in out]
    assert np.array_equal(k1, k2) and np.array_equal(_bits(d1), _bits(d2))            # idempotent
    assert np.all(c1 == k) and np.all(ec == k)                                        # complete
    assert np.all(np.diff(d1, axis=1) >= 0) and np.all(np.diff(ed, axis=1) >= 0)      # ascending
    assert np.all((k1 >= 0) & (k1 < n))
    assert all(len(set(row.tolist())) == k for row in k1)                             # no row twice
    same = 0
    for i in range(B):                                                                # same row -> same distance bits
        pos = {int(r): j for j, r in enumerate(ek[i])}
        for j, r in enumerate(k1[i]):
            if int(r) in pos:
                assert _bits(d1[i, j:j + 1])[0] == _bits(ed[i, pos[int(r)]:pos[int(r)] + 1])[0]
                same += 1
    recall = same / (B * k)
    assert recall >= 0.95, recall
    assert np.all(ed[:, 0] <= d1[:, 0])                                               # nothing beats the exact nearest


def test_build_progress_is_readable_while_building():
    """GetSinkProgress (hnsw_index_physical_create.cpp:312-327): another thread sees built_count climb to loaded_count
    while vss_build_finalize runs."""
    import threading
    n, dim = 400_000, 64
    X = datagen.mixture(n, dim, 9001, intrinsic_dim=16, basis_seed=9001, centre_scale=0.1)
    gpu = gc.gpu_index(dim, "l2sq")
    gpu.reserve(n)
    gpu.stage(np.arange(n), X)
    seen, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            seen.append(gpu.build_progress())
            time.sleep(0.002)

    import time
    t = threading.Thread(target=poll)
    t.start()
    gpu.build_finalize()
    stop.set()
    t.join()
    assert gpu.build_progress() == (n, n)
    linked = [a for a, b in seen if b == n]
    assert linked == sorted(linked) and all(0 <= a <= n for a in linked)
    assert len(set(linked)) > 3, "the poller never saw the build advance"
