"""Pins oracle/liboracle.so (the CPU restatement) to the reference.

* against committed golden vectors produced by the reference's own usearch build (always runs);
* against that build directly when oracle/_ref/libusearch_ref.so is present (authoring container, and the GPU
  box when the prebuilt .so travelled) — this also proves the committed vectors are not stale.
Bar: bit-exact (graph stream bytes, keys, f32 distance bit patterns, work counters).
"""
import os

import numpy as np
import pytest

import datagen
import golden_cases
from oracle_lib import CpuIndex

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "usearch_golden.npz")


@pytest.fixture(scope="module")
def golden():
    return dict(np.load(GOLDEN))


def _check(got, golden, prefix):
    keys = [k for k in golden if k.startswith(prefix + "/")]
    assert keys, "no golden arrays for " + prefix
    for k in keys:
        name = k[len(prefix) + 1:]
        assert name in got, k
        assert got[name].shape == golden[k].shape, k
        assert np.array_equal(got[name], golden[k]), "mismatch in " + k


@pytest.mark.parametrize("case", golden_cases.BUILD_CASES, ids=[c[0] for c in golden_cases.BUILD_CASES])
def test_build_and_search_match_reference(oracle_lib, golden, case):
    got = golden_cases.run_build_case(oracle_lib, case)
    _check(got, golden, case[0])


def test_readme_result_values(oracle_lib):
    """test/sql/hnsw/hnsw_result.test:23-28 — top-3 of [1,2,3] on the 9^3 grid has l2sq distances 0,1,1
    (array_distance 0.0, 1.0, 1.0)."""
    X = datagen.readme_grid()
    idx = CpuIndex(oracle_lib, 3, "l2sq")
    idx.reserve(len(X), 1)
    idx.add_many(np.arange(len(X)), X)
    keys, d, _ = idx.search(np.array([1, 2, 3], dtype=np.float32), 3)
    assert list(d) == [0.0, 1.0, 1.0]
    assert np.array_equal(X[keys[0]], [1, 2, 3])
    assert all(np.sum((X[k] - [1, 2, 3]) ** 2) == dd for k, dd in zip(keys, d))


def test_crud_scenario_matches_reference(oracle_lib, golden):
    _check(golden_cases.run_crud_case(oracle_lib), golden, "crud")


def test_slot_reuse_scenario_matches_reference(oracle_lib, golden):
    _check(golden_cases.run_reuse_case(oracle_lib), golden, "reuse")


def test_filtered_search_matches_reference(oracle_lib, golden):
    """usearch filtered_search.  The reference reads `top.top()` of an EMPTY buffer (index.hpp:3992, SURVEY quirk Q6 —
    undefined behaviour) whenever a rejected candidate is accepted before any admitted one; the radius becomes garbage
    and the query returns nothing.  The restatement treats the radius as unbounded until then, so rows are compared where the
    reference did not take that path (it returned k rows); elsewhere the restatement must return admitted rows."""
    got = golden_cases.run_filtered_case(oracle_lib)
    wave = golden_cases.run_filtered_case(oracle_lib, order=0, wave=1)
    n_bits = 3 * 2500 - 7
    for tag, frac, k in (("half", 0.5, 10), ("rare", 0.02, 10), ("most", 0.95, 5)):
        ref_cnt = golden["filtered/f_%s_cnt" % tag]
        ok = ref_cnt > 0  # a non-empty reference answer means its result list never was empty: no UB on that query
        assert ok.sum() >= (30 if tag != "rare" else 1), (tag, int(ok.sum()))  # rare predicate: the reference mostly returns nothing
        for part in ("keys", "dbits", "cnt", "stats"):
            name = "f_%s_%s" % (tag, part)
            assert np.array_equal(got[name][ok], golden["filtered/" + name][ok]), name
            # the kernels' variant (admitted rows in the result list, every accepted candidate in an unbounded queue) returns
            # the same rows however selective the predicate is — 2 % makes the reference walk most of the graph
            if part != "stats":
                assert np.array_equal(wave[name], got[name]), name
        bm = golden_cases.filter_bitmap(n_bits, 700 + k, frac)
        keys = got["f_%s_keys" % tag]
        live = keys[keys >= 0]
        assert np.all((bm[live >> 6] >> (live & 63).astype(np.uint64)) & np.uint64(1) == 1) and np.all(live % 33 != 0)
        assert np.all(got["f_%s_cnt" % tag][~ok] > 0)  # where the reference fell into the UB path, rows ARE found
        wk = wave["f_%s_keys" % tag]
        wl = wk[wk >= 0]
        assert np.all((bm[wl >> 6] >> (wl & 63).astype(np.uint64)) & np.uint64(1) == 1) and np.all(wl % 33 != 0)


def test_level_generator_matches_reference(oracle_lib, golden):
    _check(golden_cases.run_levels_case(oracle_lib), golden, "levels")
    for M in golden_cases.LEVEL_MS:
        out = np.zeros(1000, dtype=np.int16)
        oracle_lib.orc_draw_levels(M, 1000, out.ctypes.data)
        assert np.array_equal(out, golden["levels/levels_M%d" % M])


def test_metrics_match_reference(oracle_lib, golden):
    _check(golden_cases.run_distance_case(oracle_lib), golden, "distance")


def test_golden_is_what_the_reference_produces(ref_lib, golden):
    """Only where the reference build exists: the committed vectors are regenerated and compared."""
    fresh = golden_cases.run_all(ref_lib)
    assert set(fresh) == set(golden)
    for k in golden:
        assert np.array_equal(fresh[k], golden[k]), k


# ---- properties of the two restatement switches (see hnsw_oracle.cpp header) ----

@pytest.mark.parametrize("metric,order", [("l2sq", 0), ("l2sq", 1), ("cosine", 1), ("ip", 1)])
def test_kernel_candidate_lists_equal_reference_lists_without_ties(oracle_lib, metric, order):
    """wave=1 (the HIP kernels' single sorted list) builds the byte-identical graph and returns identical
    results as wave=0 (reference heap + sorted buffer) on tie-free data; a singleton-batch bulk build is the
    sequential add()."""
    n, d = 1500, 24
    X = datagen.mixture(n, d, 31, normalize=metric != "l2sq")
    Q = datagen.mixture(50, d, 32, n_clusters=38, normalize=metric != "l2sq")
    a = CpuIndex(oracle_lib, d, metric, 8, 16, 64, 48, order=order, wave=0)
    b = CpuIndex(oracle_lib, d, metric, 8, 16, 64, 48, order=order, wave=1)
    c = CpuIndex(oracle_lib, d, metric, 8, 16, 64, 48, order=order, wave=1)
    for ix in (a, b, c):
        ix.reserve(n, 1)
    a.add_many(np.arange(n), X)
    b.add_many(np.arange(n), X)
    c.build_batch(np.arange(n), X, 1, 1)
    assert a.save() == b.save() == c.save()
    ra, rb = a.search_many(Q, 10), b.search_many(Q, 10)
    assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1].view(np.uint32), rb[1].view(np.uint32))
    for k in range(0, n, 5):
        a.remove(k), b.remove(k)
    ra, rb = a.search_many(Q, 10, ef=30), b.search_many(Q, 10, ef=30)
    assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1].view(np.uint32), rb[1].view(np.uint32))


def test_wave_summation_order_is_within_tolerance(oracle_lib):
    """The kernels' summation tree (order=1) vs the reference's sequential sum: <= 1e-5 relative on the
    quantity being summed (north_star tolerance)."""
    for dim in golden_cases.DIST_DIMS:
        A, B = golden_cases.distance_inputs(dim)
        A, B = A[6:], B[6:]
        for mi in range(3):
            for i in range(len(A)):
                r = oracle_lib.orc_distance(mi, A[i].ctypes.data, B[i].ctypes.data, dim)
                w = oracle_lib.orc_distance_wave(mi, A[i].ctypes.data, B[i].ctypes.data, dim)
                scale = float(np.sum(np.abs(A[i] * B[i]))) if mi == 2 else abs(r)
                if mi == 1:
                    scale = 1.0
                assert abs(r - w) <= 1e-5 * max(scale, 1e-30), (dim, mi, i, r, w)


def test_batch_build_quality_and_invariants(oracle_lib):
    """The batch-synchronous build (what the GPU engine runs) keeps the reference's structural invariants and
    its recall on the same data."""
    n, d = 4000, 16
    X = datagen.mixture(n, d, 51)
    Q = datagen.mixture(100, d, 52, n_clusters=63)
    seq = CpuIndex(oracle_lib, d, "l2sq", 16, 32, 128, 64, order=1, wave=1)
    bat = CpuIndex(oracle_lib, d, "l2sq", 16, 32, 128, 64, order=1, wave=1)
    seq.reserve(n, 1), bat.reserve(n, 1)
    seq.add_many(np.arange(n), X)
    bat.build_batch(np.arange(n), X, 512, 8)
    from oracle_lib import parse_stream
    g = parse_stream(bat.save())
    assert np.array_equal(g["levels"], parse_stream(seq.save())["levels"])
    for slot, per_level in enumerate(g["adj"]):
        for lvl, nb in enumerate(per_level):
            assert len(nb) <= (32 if lvl == 0 else 16)
            assert len(set(nb.tolist())) == len(nb) and slot not in nb
            assert all(g["levels"][t] >= lvl for t in nb)
    gt = seq.search_many(Q, 10, exact=True)[0]

    def recall(ix):
        got = ix.search_many(Q, 10)[0]
        return np.mean([len(set(got[i]) & set(gt[i])) / 10 for i in range(len(Q))])
    assert recall(bat) >= recall(seq) - 0.02


@pytest.mark.parametrize("order,wave", [(0, 0), (1, 1)])
def test_batch_build_with_reuse_equals_sequential_adds(oracle_lib, ref_lib, order, wave):
    """build_batch with singleton batches over an index with tombstones takes the reference's update() path row by row:
    same slots reused in the same (free-ring) order, same lists, same stream as a loop of add() calls — and, in
    reference order, the same bytes as the reference library itself."""
    d = 12
    X = datagen.mixture(900, d, 4242)
    Q = datagen.mixture(30, d, 4243, n_clusters=24)

    def scenario(lib, batched, **mode):
        ix = CpuIndex(lib, d, "l2sq", 8, 16, 40, 30, **mode)
        ix.reserve(1024)
        add = (lambda k, v: ix.build_batch(k, v, 1, 1)) if batched else ix.add_many
        add(np.arange(300), X[:300])
        for k in range(10, 130, 3):
            ix.remove(k)
        add(np.arange(300, 350), X[300:350])
        for k in range(140, 290, 2):   # 75 pushes into the (empty again) 64-entry ring: it wraps (quirk Q11)
            ix.remove(k)
        add(np.arange(350, 520), X[350:520])
        return ix.save(), ix.search_many(Q, 5, ef=40)[0]

    seq = scenario(oracle_lib, False, order=order, wave=wave)
    bat = scenario(oracle_lib, True, order=order, wave=wave)
    assert seq[0] == bat[0] and np.array_equal(seq[1], bat[1])
    if ref_lib is not None and (order, wave) == (0, 0):
        ref = scenario(ref_lib, False)
        assert ref[0] == bat[0] and np.array_equal(ref[1], bat[1])


@pytest.mark.parametrize("seed", range(12))
def test_random_scripts_match_reference(oracle_lib, ref_lib, seed):
    """Differential test, restatement vs the reference library, over random op sequences: add / remove (present, absent,
    repeated) / search at random ef / exact search / reserve growth / save + reload — after every step the
    return values agree, at checkpoints the serialized streams are byte-identical.  (Only where the reference build
    exists; the golden vectors cover the rest.)  compact() is left out: on an index with tombstones the reference's own
    compact corrupts its heap ("invalid fastbin entry (free)" after 70 random ops, seed 0) — the controlled compact
    scenario of the CRUD golden is what pins it."""
    if ref_lib is None:
        pytest.skip("reference build not present")
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.choice([3, 8, 17, 64]))
    metric = ["l2sq", "cosine", "ip"][seed % 3]
    M = int(rng.choice([4, 8, 16]))
    X = datagen.mixture(3000, d, 7000 + seed, normalize=metric != "l2sq")
    Q = datagen.mixture(64, d, 8000 + seed, n_clusters=20, normalize=metric != "l2sq")
    a = CpuIndex(oracle_lib, d, metric, M, 2 * M, 48, 32)
    b = CpuIndex(ref_lib, d, metric, M, 2 * M, 48, 32)
    cap = 64
    a.reserve(cap), b.reserve(cap)
    alive, next_key, reloaded = [], 0, False
    for step in range(1500):
        op = rng.random()
        if op < 0.55 or not alive:                                    # add
            if a.nodes() + 1 > cap:
                cap *= 2
                a.reserve(cap), b.reserve(cap)
            sa, sb = a.add(next_key, X[next_key % len(X)]), b.add(next_key, X[next_key % len(X)])
            assert sa.tolist() == sb.tolist(), (step, "add stats/slot")
            alive.append(next_key)
            next_key += 1
        elif op < 0.72:                                               # remove: present / absent / already removed
            k = alive.pop(int(rng.integers(len(alive)))) if rng.random() < 0.8 else int(rng.integers(0, next_key + 5))
            assert a.remove(k) == b.remove(k), (step, "remove")
            if k in alive:
                alive.remove(k)
        elif op < 0.93:                                               # search
            q = Q[int(rng.integers(len(Q)))]
            kk, ef, exact = int(rng.choice([1, 5, 10])), int(rng.choice([4, 16, 40, 100])), bool(rng.random() < 0.15)
            ka, da, sa = a.search(q, kk, ef=ef, exact=exact)
            kb, db, sb = b.search(q, kk, ef=ef, exact=exact)
            assert ka.tolist() == kb.tolist(), (step, "search keys")
            assert da.view(np.uint32).tolist() == db.view(np.uint32).tolist(), (step, "search distance bits")
            assert sa.tolist() == sb.tolist(), (step, "search counters")
        else:                                                         # checkpoint: streams, sizes; sometimes reload
            assert (a.size(), a.nodes(), a.capacity(), a.max_level()) == (b.size(), b.nodes(), b.capacity(), b.max_level())
            blob_a, blob_b = a.save(), b.save()
            assert blob_a == blob_b, (step, "stream")
            if rng.random() < 0.3:
                a = CpuIndex(oracle_lib, d, metric, M, 2 * M, 48, 32)
                b = CpuIndex(ref_lib, d, metric, M, 2 * M, 48, 32)
                a.load(blob_a), b.load(blob_b)
                cap = max(cap, a.capacity())
                a.reserve(cap), b.reserve(cap)
                reloaded = True   # after a reload the reference forgets its key map (quirk Q4): removes become no-ops
    assert a.save() == b.save()


@pytest.mark.parametrize("seed", range(10))
def test_random_chunked_inserts_with_reuse_match_reference(oracle_lib, ref_lib, seed):
    """The batch entry point the GPU engine mirrors (build_batch), driven with singleton batches in random chunk sizes
    between random deletions, against the reference's add()/remove() calls: same slots reused, same streams."""
    if ref_lib is None:
        pytest.skip("reference build not present")
    rng = np.random.default_rng(300 + seed)
    o = _random_options(rng)
    d, metric = max(2, o["d"]), o["metric"]
    X = datagen.mixture(4000, d, 9100 + seed, normalize=metric != "l2sq")
    a = CpuIndex(oracle_lib, d, metric, o["M"], o["M0"], o["efc"], o["efs"])
    b = CpuIndex(ref_lib, d, metric, o["M"], o["M0"], o["efc"], o["efs"])
    a.reserve(4096), b.reserve(4096)
    alive, key = [], 0
    for round_ in range(40):
        n = int(rng.integers(1, 120))
        keys = np.arange(key, key + n)
        a.build_batch(keys, X[key:key + n], 1, 1)
        b.add_many(keys, X[key:key + n])
        alive += keys.tolist()
        key += n
        for _ in range(int(rng.integers(0, 90))):
            if not alive:
                break
            k = alive.pop(int(rng.integers(len(alive))))
            assert a.remove(k) == b.remove(k) == 1
        assert a.nodes() == b.nodes()
        assert a.save() == b.save(), round_


def _random_options(rng):
    M = int(rng.integers(2, 21))
    return dict(d=int(rng.choice([1, 2, 3, 5, 16, 33, 100])), metric=["l2sq", "cosine", "ip"][int(rng.integers(3))], M=M,
                M0=int(rng.integers(M, 41)), efc=int(rng.integers(1, 81)), efs=int(rng.integers(1, 81)))


@pytest.mark.parametrize("seed", range(16))
def test_random_option_space_matches_reference(oracle_lib, ref_lib, seed):
    """The whole option space — M 2..20, M0 M..40 (M0 < M overruns the reference's own lists), ef_construction and
    ef_search 1..80, dimensions 1..100, all metrics, k above ef, data with exact ties and zero vectors — 500 random
    add / remove / search / exact-search / save ops per configuration: every return value and every stream byte equal."""
    if ref_lib is None:
        pytest.skip("reference build not present")
    rng = np.random.default_rng(50_000 + seed)
    o = _random_options(rng)
    d, metric, n = o["d"], o["metric"], 1200
    kind = int(rng.integers(3))
    if kind == 0:
        X = datagen.mixture(n, d, seed, normalize=metric != "l2sq")
    elif kind == 1:
        X = rng.integers(0, 4, size=(n, d)).astype(np.float32)        # a small integer grid: ties everywhere
    else:
        X = datagen.mixture(n, d, seed)
        X[rng.integers(0, n, size=20)] = 0                             # zero vectors (cosine's special cases)
    Q = np.concatenate([X[rng.integers(0, n, size=16)], datagen.mixture(16, d, seed + 1)]).astype(np.float32)
    a = CpuIndex(oracle_lib, d, metric, o["M"], o["M0"], o["efc"], o["efs"])
    b = CpuIndex(ref_lib, d, metric, o["M"], o["M0"], o["efc"], o["efs"])
    cap = 32
    a.reserve(cap), b.reserve(cap)
    alive, key = [], 0
    for step in range(500):
        op = rng.random()
        if op < 0.6 or not alive:
            if a.nodes() + 1 > cap:
                cap *= 2
                a.reserve(cap), b.reserve(cap)
            assert a.add(key, X[key % n]).tolist() == b.add(key, X[key % n]).tolist(), (o, step, "add")
            alive.append(key)
            key += 1
        elif op < 0.72:
            k = alive.pop(int(rng.integers(len(alive))))
            assert a.remove(k) == b.remove(k), (o, step, "remove")
        elif op < 0.95:
            q = Q[int(rng.integers(len(Q)))]
            kk, ef, exact = int(rng.choice([1, 3, 10, 50])), int(rng.choice([1, 2, 8, 30, 100])), bool(rng.random() < 0.1)
            ka, da, sa = a.search(q, kk, ef=ef, exact=exact)
            kb, db, sb = b.search(q, kk, ef=ef, exact=exact)
            assert ka.tolist() == kb.tolist(), (o, step, "keys")
            assert da.view(np.uint32).tolist() == db.view(np.uint32).tolist(), (o, step, "distance bits")
            assert sa.tolist() == sb.tolist(), (o, step, "counters")
        else:
            assert a.save() == b.save(), (o, step, "stream")
            assert (a.size(), a.nodes(), a.capacity(), a.max_level()) == (b.size(), b.nodes(), b.capacity(), b.max_level())
            for level in range(4):
                assert a.level_stats(level).tolist() == b.level_stats(level).tolist(), (o, step, "stats", level)
    assert a.save() == b.save(), o


@pytest.mark.parametrize("seed", range(12))
def test_random_option_space_kernel_lists_equal_reference_lists(oracle_lib, seed):
    """The kernels' candidate structure (one sorted list with expanded marks; two lists once tombstones exist) against the
    reference's heap + sorted buffer over the same random option space, on tie-free data, deletions included: identical
    graphs and answers."""
    rng = np.random.default_rng(60_000 + seed)
    o = _random_options(rng)
    d, metric, n = max(3, o["d"]), o["metric"], 1200
    X = datagen.mixture(n, d, seed, normalize=metric != "l2sq")
    Q = datagen.mixture(32, d, seed + 1, normalize=metric != "l2sq")
    a = CpuIndex(oracle_lib, d, metric, o["M"], o["M0"], o["efc"], o["efs"], order=0, wave=0)
    b = CpuIndex(oracle_lib, d, metric, o["M"], o["M0"], o["efc"], o["efs"], order=0, wave=1)
    a.reserve(2048), b.reserve(2048)
    alive, key = [], 0
    for step in range(500):
        op = rng.random()
        if op < 0.6 or not alive:
            a.add(key, X[key % n]), b.add(key, X[key % n])
            alive.append(key)
            key += 1
        elif op < 0.70:
            k = alive.pop(int(rng.integers(len(alive))))
            assert a.remove(k) == b.remove(k)
        elif op < 0.95:
            q = Q[int(rng.integers(len(Q)))]
            kk, ef = int(rng.choice([1, 3, 10, 50])), int(rng.choice([1, 2, 8, 30, 100]))
            ka, da, _ = a.search(q, kk, ef=ef)
            kb, db, _ = b.search(q, kk, ef=ef)
            assert ka.tolist() == kb.tolist() and da.view(np.uint32).tolist() == db.view(np.uint32).tolist(), (o, step)
        else:
            assert a.save() == b.save(), (o, step)
    assert a.save() == b.save(), o


@pytest.mark.parametrize("metric", ["l2sq", "cosine"])
def test_kernel_lists_equal_reference_lists_beyond_512_and_under_rare_predicates(oracle_lib, metric):
    """The limits the reference accepts without bound (LIMIT k on the scan path, hnsw_optimize_scan.cpp:146; k < 2048 on
    the top-k path, hnsw_optimize_topk.cpp:170-173; any ef_search / ef_construction >= 1, hnsw_index_plan.cpp:33-80):
    the kernels' lists (wave=1) give the reference's answers (wave=0) for k in {600, 2000}, ef_search 1024, an
    ef_construction of 700, and a 2 % predicate with tombstones, on tie-free data."""
    n, d = 3000, 16
    X = datagen.mixture(n, d, 77, normalize=metric != "l2sq")
    Q = datagen.mixture(12, d, 78, n_clusters=50, normalize=metric != "l2sq")
    a = CpuIndex(oracle_lib, d, metric, 8, 16, 700, 64, order=1, wave=0)
    b = CpuIndex(oracle_lib, d, metric, 8, 16, 700, 64, order=1, wave=1)
    for ix in (a, b):
        ix.reserve(n, 1)
    a.add_many(np.arange(n), X)
    b.build_batch(np.arange(n), X, 1, 1)
    assert a.save() == b.save()
    for k, ef in ((600, 64), (2000, 100), (10, 1024), (3500, 5000)):
        ra, rb = a.search_many(Q, k, ef=ef), b.search_many(Q, k, ef=ef)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1].view(np.uint32), rb[1].view(np.uint32))
        assert np.array_equal(ra[2], rb[2]) and np.array_equal(ra[3], rb[3])
    for key in range(0, n, 7):
        a.remove(key), b.remove(key)
    bm = golden_cases.filter_bitmap(n, 5, 0.02)
    for k, ef in ((10, 64), (40, 100), (600, 1024)):
        ra, rb = a.search_many_filtered(Q, k, ef, bm, n), b.search_many_filtered(Q, k, ef, bm, n)
        ok = ra[2] > 0  # the reference's own empty-buffer read (quirk Q6) returns nothing on some queries
        assert np.array_equal(ra[0][ok], rb[0][ok]) and np.array_equal(ra[2][ok], rb[2][ok])
        assert np.all(rb[2] > 0)


@pytest.mark.parametrize("ties", [False, True])
def test_register_queue_of_pending_candidates_never_changes_an_answer(oracle_lib, ties):
    """The engine keeps the pending candidates of a search over tombstones / a predicate in a register queue of twice the
    result list's size (RegQueue) and forgets the farthest one when it is full ONLY if the result list is full and that
    candidate lies beyond the radius; otherwise the query overflows and is re-run with the unbounded queue (usearch's
    `next` heap, index.hpp:3981-3992).  Modelled in the oracle's kernel mode: over predicates from 90 % to 5 %, with
    tombstones, with and without distance ties, and for queues much smaller than the engine's (so that the rule is
    exercised thousands of times), every query that did not overflow returns exactly the unbounded answer, work counters
    included; with the engine's size, mild predicates almost never overflow."""
    n, d = 4000, 20
    X = datagen.mixture(n, d, 9090)
    Q = datagen.mixture(48, d, 9091, n_clusters=40)
    if ties:  # coarse lattice: equal distances everywhere, duplicated rows
        X, Q = np.rint(X * 1.5).astype(np.float32), np.rint(Q * 1.5).astype(np.float32)
    idx = CpuIndex(oracle_lib, d, "l2sq", 12, 24, 100, 64, order=1, wave=1)
    idx.reserve(n, 1)
    idx.build_batch(np.arange(n), X, 64, 8)
    for key in range(3, n, 19):
        idx.remove(key)

    def run(q, k, ef, bm):
        if bm is None:
            r = idx.search_many(q[None, :], k, ef=ef)
        else:
            r = idx.search_many_filtered(q[None, :], k, ef, bm, n)
        return r[0][0].tolist(), r[1][0].view(np.uint32).tolist(), int(r[2][0]), r[3][0].tolist()

    drops_total, checked, fast_engine_size = 0, 0, {}
    for frac, k, ef in ((None, 10, 64), (0.9, 10, 64), (0.5, 10, 64), (0.5, 5, 20), (0.2, 10, 100), (0.05, 10, 40), (0.9, 100, 200)):
        bm = None if frac is None else golden_cases.filter_bitmap(n, 11, frac)
        limit = max(k, ef)
        engine_cap = 2 * 64 * (2 if limit <= 128 else 4)
        for cap in (engine_cap, limit, 48, 16):
            fast = 0
            for q in Q:
                idx.set_register_queue(0)
                want = run(q, k, ef, bm)
                idx.set_register_queue(cap)
                got = run(q, k, ef, bm)
                overflowed, drops = idx.register_queue_state()
                if not overflowed:
                    assert got == want, (frac, k, ef, cap)
                    fast += 1
                    checked += 1
                    drops_total += drops
            if cap == engine_cap:
                fast_engine_size[(frac, k, ef)] = fast / len(Q)
    idx.set_register_queue(0)
    assert checked > 300 and drops_total > 1000, (checked, drops_total)
    for case in ((None, 10, 64), (0.9, 10, 64), (0.9, 100, 200)):  # tombstones only / a mild predicate: the register queue suffices
        assert fast_engine_size[case] >= 0.9, fast_engine_size


@pytest.mark.parametrize("metric", ["l2sq", "cosine"])
def test_engine_compaction_order_is_the_reference_compaction_order(oracle_lib, ref_lib, metric):
    """vss_compact's CPU mirror (compact_reordering) against index_gt::compact of the REFERENCE LIBRARY (index.hpp:3405-3494)
    on an index without tombstones (where the reference's compact is sound, quirk Q3): the reference sorts by (level
    descending, cluster ascending) with std::sort, the mirror with a stable sort, so the two numberings may differ only
    inside a group of equal (level, cluster).  Compared through what the numbering cannot hide: along the new slots the
    levels are identical, the multiset of keys inside every run of equal (level, cluster) is identical (the cluster of a
    run is named by the KEY of the landing node), every node keeps its neighbour lists (as key lists, in order), the entry
    is the same key — and every search returns the same keys with the same distance bits."""
    from oracle_lib import parse_stream
    n, dim = 2500, 24
    X = datagen.mixture(n, dim, 4242, normalize=metric != "l2sq")
    Q = datagen.mixture(50, dim, 4243, n_clusters=50, normalize=metric != "l2sq")
    ref, orc = CpuIndex(ref_lib, dim, metric, 8, 16, 40), CpuIndex(oracle_lib, dim, metric, 8, 16, 40)
    ref.reserve(n), orc.reserve(n)
    keys = np.arange(n) * 7 + 3
    ref.add_many(keys, X), orc.add_many(keys, X)
    assert ref.save() == orc.save()
    before = orc.search_many(Q, 10, ef=50)
    # clusters by KEY, taken before anything moves: the landing node of each row's descent (search_for_one_ ... level 0)
    g0 = parse_stream(orc.save())
    ref.compact()
    orc.compact_reordering()
    a, b = parse_stream(ref.save()), parse_stream(orc.save())
    assert np.array_equal(a["levels"], b["levels"]) and np.all(np.diff(b["levels"].astype(np.int32)) <= 0)
    assert a["keys"][a["entry"]] == b["keys"][b["entry"]] and a["max_level"] == b["max_level"] == g0["max_level"]
    assert sorted(a["keys"].tolist()) == sorted(b["keys"].tolist()) == sorted(keys.tolist())
    # same lists per node, as keys
    def by_key(g):
        return {int(g["keys"][s]): [g["keys"][nb].tolist() for nb in g["adj"][s]] for s in range(g["rows"])}
    la, lb, l0 = by_key(a), by_key(b), by_key(g0)
    assert la == lb == l0
    # the two numberings agree up to the order inside groups of equal (level, cluster): cut the reference's numbering into
    # maximal runs that the mirror fills with the same key sets at the same positions
    pos_b = {int(k): s for s, k in enumerate(b["keys"])}
    s = 0
    groups = 0
    while s < n:
        lo = hi = pos_b[int(a["keys"][s])]
        e = s + 1
        seen = {lo}
        # grow the run until the reference's slots [s, e) and the mirror's slots cover the same interval
        while not (lo == s and hi == e - 1 and len(seen) == e - s):
            p = pos_b[int(a["keys"][e])] if e < n else None
            assert p is not None, "numberings differ by more than the order inside (level, cluster) groups"
            seen.add(p)
            lo, hi = min(lo, p), max(hi, p)
            e += 1
        assert len({int(a["levels"][t]) for t in range(s, e)}) == 1  # a run never straddles a level
        groups += 1
        s = e
    assert groups > n // 8  # the runs are small (clusters), not one big permutation
    for idx in (ref, orc):
        k2, d2, c2, _ = idx.search_many(Q, 10, ef=50)
        assert np.array_equal(k2, before[0]) and np.array_equal(d2.view(np.uint32), before[1].view(np.uint32))


def test_engine_compaction_with_tombstones_prunes_and_reorders(oracle_lib):
    """With tombstones (where the reference's own compact is unsound): compact_reordering == compact_dropping followed by a
    pure renumbering — same keys, same per-node key lists, same answers — and the dropped keys are gone."""
    from oracle_lib import parse_stream
    n, dim = 2000, 16
    X = datagen.mixture(n, dim, 777)
    Q = datagen.mixture(40, dim, 778, n_clusters=44)
    pair = []
    for _ in range(2):
        idx = CpuIndex(oracle_lib, dim, "l2sq", 8, 16, 48, order=1, wave=1)
        idx.reserve(n)
        idx.build_batch(np.arange(n), X, 128, 8)
        pair.append(idx)
    rng = np.random.default_rng(9)
    dead = sorted(set(rng.choice(n, 400, replace=False).tolist() + [int(pair[0].entry_slot())]))
    for idx in pair:
        for r in dead:
            idx.remove(r)
    pair[0].compact_dropping()
    pair[1].compact_reordering()
    a, b = parse_stream(pair[0].save()), parse_stream(pair[1].save())
    assert sorted(a["keys"].tolist()) == sorted(b["keys"].tolist()) and not set(b["keys"].tolist()) & set(dead)
    assert np.all(np.diff(b["levels"].astype(np.int32)) <= 0)

    def by_key(g):
        return {int(g["keys"][s]): [g["keys"][nb].tolist() for nb in g["adj"][s]] for s in range(g["rows"])}
    assert by_key(a) == by_key(b)
    ka, da, _, _ = pair[0].search_many(Q, 10, ef=64)
    kb, db, _, _ = pair[1].search_many(Q, 10, ef=64)
    assert np.array_equal(ka, kb) and np.array_equal(da.view(np.uint32), db.view(np.uint32))
    assert pair[1].size() == pair[1].nodes() == n - len(dead)


@pytest.mark.parametrize("metric,ties", [("l2sq", False), ("cosine", False), ("ip", False), ("l2sq", True)])
def test_the_successor_of_an_expansion_is_told_by_its_fresh_scores(oracle_lib, metric, ties):
    """Round 4: the engine's software-pipelined level search (hnsw_kernels.h, level_search_pipelined) picks the candidate it
    expands NEXT from an expansion's fresh scores before inserting any of them — m = the smallest fresh distance the radius
    test admits, e = the best unexpanded entry, next = row(m) if (no e or m < e.distance) else e — and only then runs the
    sorted inserts, in the shadow of the successor's row loads.  The rule is claimed exact whenever m is not exactly tied
    (with another admitted fresh row, with a list entry) and no NaN is involved; such expansions take the plain order.
    Modelled in the oracle's kernel mode and checked against the plain order on every expansion: over three metrics, small and
    large limits, lists that are filling and lists that are full — zero wrong predictions; on generic data fewer than one
    expansion in a thousand is left to the plain order; on a coarse lattice (ties everywhere, duplicated rows) many are, and the rest are
    still predicted right.  The answers themselves are untouched by the check."""
    n, d = 5000, 24
    X = datagen.mixture(n, d, 4711)
    Q = datagen.mixture(64, d, 4712, n_clusters=40)
    if ties:
        X, Q = np.rint(X * 1.5).astype(np.float32), np.rint(Q * 1.5).astype(np.float32)
    if metric != "l2sq":
        X /= np.maximum(np.linalg.norm(X, axis=1, keepdims=True), 1e-9)
        Q /= np.maximum(np.linalg.norm(Q, axis=1, keepdims=True), 1e-9)
    idx = CpuIndex(oracle_lib, d, metric, 16, 32, 80, 64, order=1, wave=1)
    idx.reserve(n, 1)
    idx.build_batch(np.arange(n), X, 128, 8)
    checked = left = 0
    for k, ef in ((10, 16), (10, 64), (3, 200), (100, 256)):
        idx.set_pipeline_check(False)
        want = idx.search_many(Q, k, ef=ef)
        idx.set_pipeline_check(True)
        got = idx.search_many(Q, k, ef=ef)
        c, t, wrong = idx.pipeline_check_state()
        assert wrong == 0, (metric, ties, k, ef, c, t, wrong)
        assert c > len(Q) * 5  # every level-0 expansion of every query went through the check
        for a, b in zip(want, got):  # the check changes nothing: ids, distance bits, counts, work counters
            assert np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))
        checked, left = checked + c, left + t
    idx.set_pipeline_check(False)
    if ties:
        assert 0 < left < checked            # exact ties do occur and take the plain order; most expansions are still predicted
    else:                                    # generic data: (almost) every expansion's successor is told from its fresh scores —
        assert left <= checked // 1000       # an exact f32 tie is a rare accident (1 - a.b of unit vectors rounds to few values)
