"""CPU tests of the engine's pure host logic (duckdb-vss_amd/csrc/host_logic.h, the very header libvssgpu.so is built
from): level generator, batch schedule, free-slot ring, rowid map — against the oracle and the reference build."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import datagen
from oracle_lib import CpuIndex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hl():
    src = os.path.join(HERE, "host_logic_probe.cpp")
    hdrs = [os.path.join(ROOT, "duckdb-vss_amd", "csrc", h) for h in ("host_logic.h", "visited_compact.h")]
    out = os.path.join(HERE, "host_logic_probe.so")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", src, "-o", out])
    lib = C.CDLL(out)
    lib.hl_schedule.restype = C.c_uint64
    lib.hl_keymap_check.restype = C.c_uint64
    return lib


@pytest.mark.parametrize("M", [2, 3, 16, 32, 48])
def test_engine_level_generator_is_the_reference_generator(hl, oracle_lib, golden_levels, M):
    n = 200_000
    got = np.zeros(n, dtype=np.uint8)
    hl.hl_draw_levels(C.c_uint64(M), C.c_uint64(n), got.ctypes.data_as(C.c_void_p))
    want = np.zeros(n, dtype=np.int16)
    oracle_lib.orc_draw_levels(M, n, want.ctypes.data)
    assert np.array_equal(got.astype(np.int16), want)
    key = "levels/levels_M%d" % M
    if key in golden_levels:  # the first 1000 draws as the reference library itself produced them
        assert np.array_equal(got[:1000].astype(np.int16), golden_levels[key])


@pytest.fixture(scope="module")
def golden_levels():
    return dict(np.load(os.path.join(HERE, "golden", "usearch_golden.npz")))


@pytest.mark.parametrize("existing,max_batch,growth_div", [(0, 16384, 32), (0, 1, 1), (5000, 256, 8), (100, 64, 4),
                                                           (1_000_000, 16384, 32)])
def test_engine_batch_schedule_equals_the_oracle_schedule(hl, oracle_lib, existing, max_batch, growth_div):
    n = 60_000
    lv = np.zeros(n, dtype=np.int16)
    oracle_lib.orc_draw_levels(16, n, lv.ctypes.data)
    lv = np.roll(lv, 7)                      # not the generator's own order: promotions at arbitrary places
    cur_max = -1 if existing == 0 else 3
    want = np.zeros(n, dtype=np.uint64)
    nw = oracle_lib.orc_schedule(existing, cur_max, lv.ctypes.data, n, max_batch, growth_div, want.ctypes.data)
    got = np.zeros(n, dtype=np.uint64)
    ng = hl.hl_schedule(C.c_uint64(existing), cur_max, lv.astype(np.uint8).ctypes.data_as(C.c_void_p), C.c_uint64(n),
                        C.c_uint64(max_batch), C.c_uint64(growth_div), C.c_int64(-1), got.ctypes.data_as(C.c_void_p))
    assert ng == nw and np.array_equal(got[:ng], want[:nw])
    assert int(got[:ng].sum()) == n
    # a row that re-links the entry slot runs alone
    solo = 12_345
    ns = hl.hl_schedule(C.c_uint64(max(existing, 50_000)), 9, lv.astype(np.uint8).ctypes.data_as(C.c_void_p), C.c_uint64(n),
                        C.c_uint64(max_batch), C.c_uint64(growth_div), C.c_int64(solo), got.ctypes.data_as(C.c_void_p))
    ends = np.cumsum(got[:ns])
    i = int(np.searchsorted(ends, solo, side="right"))
    assert got[i] == 1 and (ends[i] - 1) == solo


def test_engine_free_ring_hands_out_the_slots_the_reference_hands_out(hl, ref_lib):
    """Removes and inserts interleaved so that the 64-entry ring wraps (quirk Q11): the slot every insert lands in, as
    the reference library reports it (add_result_t::slot), is what the engine's ring hands out."""
    if ref_lib is None:
        pytest.skip("reference build not present")
    d = 8
    X = datagen.mixture(1200, d, 99)
    idx = CpuIndex(ref_lib, d, "l2sq", 8, 16, 32, 32)
    idx.reserve(2048, 1)
    idx.add_many(np.arange(400), X[:400])
    slot_of = {k: k for k in range(400)}     # sequential adds into an empty index: slot = position
    nodes, key = 400, 400
    ops, want = [], []
    rng = np.random.default_rng(5)
    alive = list(range(400))
    for n_del, n_add in [(30, 10), (90, 40), (5, 100), (130, 200), (64, 64), (65, 70)]:
        for _ in range(n_del):
            k = alive.pop(int(rng.integers(len(alive))))
            assert idx.remove(k) == 1
            ops.append(slot_of.pop(k) + 1)
            want.append(-2)
        for _ in range(n_add):
            slot = int(idx.add(key, X[key % len(X)])[2])
            ops.append(0)
            if slot == nodes:                # the reference appended: its ring had nothing to offer
                want.append(-1)
                nodes += 1
            else:
                want.append(slot)
            slot_of[key] = slot
            alive.append(key)
            key += 1
    ops = np.array(ops, dtype=np.int64)
    got = np.zeros(len(ops), dtype=np.int64)
    hl.hl_ring_script(ops.ctypes.data_as(C.c_void_p), C.c_uint64(len(ops)), got.ctypes.data_as(C.c_void_p))
    assert got.tolist() == want
    assert any(w >= 0 for w in want) and any(w == -1 for w in want)


def test_engine_rowid_map(hl):
    rng = np.random.default_rng(11)
    keys = rng.permutation(200_000).astype(np.int64) * 7 - 300_000      # negative and positive row ids
    erase = (rng.random(len(keys)) < 0.3).astype(np.uint8)
    assert hl.hl_keymap_check(keys.ctypes.data_as(C.c_void_p), erase.ctypes.data_as(C.c_void_p), C.c_uint64(len(keys))) == 0


def test_batched_schedule_builds_a_graph_as_good_as_the_reference_build():
    """profiles/r02_build_quality.json (tools/study_build_quality.py, 1M rows x 128, reference default options): recall@10
    against exact brute force of the graph the engine's batch-synchronous schedule builds (the CPU restatement in kernel
    mode = the GPU's graph byte for byte) vs the reference library's own multi-stream build, same data and queries.
    Batch mates do not see each other during a batch: the cost is at most 1.5 recall points on the hardest data spec."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_build_quality.json")
    rows = json.load(open(path))["results"]
    assert {r["centre_scale"] for r in rows} == {0.1, 1.0} and all(r["rows"] == 1_000_000 for r in rows)
    for r in rows:
        if r.get("growth_div", 32) == 32:  # the engine's schedule
            for ef in (64, 128):
                assert r["batched_recall_ef%d" % ef] >= r["reference_recall_ef%d" % ef] - 0.015, (r["centre_scale"], ef)
    # the reference's multi-threaded build is a race (slot = fetch_add): its own recall moves by a point between runs; every
    # batched graph of the hard data spec, whatever the batch growth, stays within a point of the reference's worst run
    hard = [r for r in rows if r["centre_scale"] == 0.1]
    assert len(hard) >= 3
    for ef in (64, 128):
        worst_reference = min(r["reference_recall_ef%d" % ef] for r in hard)
        assert all(r["batched_recall_ef%d" % ef] >= worst_reference - 0.01 for r in hard), ef


def _shape(hl, n, dim, M0, solo_lds=70 * 1024, solo_mode=1, team=True, touch_rows=True, touch_lists=True, n_cus=256,
           force_looping=False, crew=True, engine_walkers=0, full=False):
    V = (dim + 3) // 4
    G = 1
    while G < min(64, V):
        G *= 2
    out = (C.c_uint32 * 6)()
    hl.hl_search_shape(C.c_uint32(n), C.c_uint64(M0), C.c_uint64(V), C.c_uint64(G), C.c_uint32(solo_lds), C.c_uint32(solo_mode),
                       int(team), int(touch_rows), int(touch_lists), C.c_uint32(n_cus), int(force_looping), int(crew),
                       C.c_uint32(engine_walkers), out)
    assert out[0] == out[3]  # the engine asks wants_solo() first (to size the visited set), then choose_search_shape()
    if full:  # (solo, team, touches, crew, roomy visited set)
        return bool(out[0]), bool(out[1]), int(out[2]), bool(out[4]), bool(out[5])
    return bool(out[0]), bool(out[1]), int(out[2])


def test_search_shape_policy(hl):
    """Which shape of the search engine answers a launch: the one-query probe of HNSW_INDEX_SCAN and the
    <= 204-query chunks of HNSW_INDEX_JOIN (floor(2048 / k), hnsw_optimize_join.cpp:111-168) over narrow rows run as teams
    with both touches; beyond one query per compute unit, and for wide rows, the workgroup engine — whose last walker per
    workgroup runs the scoring waves as a crew (round 4), from the first expansion on (with ListTouch and a roomy visited set)
    when the launch has at most one query per compute unit; the setters override."""
    LISTS = 0x100
    # reference defaults at 128 dims (M0 = 32: a level-0 list of rows = 16 KiB): teams up to one query per compute unit
    for n in (1, 8, 32, 204, 256):
        assert _shape(hl, n, 128, 32) == (True, True, 4 | LISTS), n
    assert _shape(hl, 257, 128, 32) == (False, False, 0)
    assert _shape(hl, 1024, 128, 32) == (False, False, 0)
    # M = 32 at 128 dims: 64 rows x 512 B = 32 KiB, still narrow; at 256 dims (8 lines per row) the helpers touch all 8
    assert _shape(hl, 1, 128, 64) == (True, True, 4 | LISTS)
    assert _shape(hl, 1, 256, 32) == (True, True, 8 | LISTS)
    assert _shape(hl, 1, 256, 64) == (False, False, LISTS)  # 64 KiB of rows per expansion: the engine's scoring waves (as a crew)
    # wide rows (the headline index, 768 dims) stay with the workgroup engine whatever the batch size; up to one query per
    # compute unit its single walker per workgroup is a latency chain (crew from the start, ListTouch, 64-KiB visited set),
    # beyond that the crew only takes over in the drain
    for n in (1, 32, 204, 256):
        assert _shape(hl, n, 768, 64, full=True) == (False, False, LISTS, True, True), n
    for n in (257, 1024, 10240):
        assert _shape(hl, n, 768, 64, full=True) == (False, False, 0, True, False), n
    assert _shape(hl, 1, 768, 64, crew=False, full=True) == (False, False, 0, False, True)
    assert _shape(hl, 1, 768, 64, touch_lists=False, full=True) == (False, False, 0, True, True)
    assert _shape(hl, 1, 768, 64, engine_walkers=4, full=True) == (False, False, 0, True, False)  # walkers forced (A/B)
    assert _shape(hl, 1024, 768, 64, engine_walkers=1, full=True) == (False, False, 0, True, True)
    # dimensions that do not fill a lane group take the looping kernels, which have a team variant; 3 dims = one line
    assert _shape(hl, 1, 96, 32) == (True, True, 3 | LISTS)
    assert _shape(hl, 1, 3, 32) == (True, True, 1 | LISTS)
    # teams off: the one-wave shape up to 32 queries, its own RowTouch only for rows of at most four lines
    assert _shape(hl, 32, 128, 32, team=False) == (True, False, 4 | LISTS)
    assert _shape(hl, 33, 128, 32, team=False) == (False, False, LISTS)
    assert _shape(hl, 1, 256, 32, team=False) == (True, False, LISTS)
    # forced shapes (vss_set_search_solo 0 / 2): never / always; a forced solo launch of more queries than compute units
    # runs one wave per query and, beyond the touch threshold, touches nothing
    assert _shape(hl, 1, 128, 32, solo_mode=0) == (False, False, LISTS)
    assert _shape(hl, 1, 128, 32, solo_mode=0, crew=False) == (False, False, 0)
    assert _shape(hl, 700, 128, 32, solo_mode=2) == (True, False, 0)
    assert _shape(hl, 200, 768, 64, solo_mode=2) == (True, False, LISTS)  # no team variant for 3 chunks per lane, 24 lines
    # the touches can be switched off one by one; a workgroup whose LDS has no room for the team's box runs alone
    assert _shape(hl, 1, 128, 32, touch_rows=False) == (True, True, LISTS)
    assert _shape(hl, 1, 128, 32, touch_lists=False) == (True, True, 4)
    assert _shape(hl, 1, 128, 32, solo_lds=160 * 1024 - 100) == (True, False, 4 | LISTS)
    # a smaller device: a team wants a compute unit per query
    assert _shape(hl, 100, 128, 32, n_cus=64) == (False, False, 0)
    assert _shape(hl, 64, 128, 32, n_cus=64) == (True, True, 4 | LISTS)


def test_batch_schedule_invariants_on_random_level_sequences(hl):
    """Properties the batch-synchronous build relies on, over random level sequences and options: the batches tile the rows in
    order; none exceeds max_batch or nodes / growth_div; a row above the running top level is a batch of its own (it becomes
    the entry, index.hpp:2769-2772) and only then does the top level rise; the very first row of an empty index runs alone."""
    rng = np.random.default_rng(20260927)
    for trial in range(60):
        n = int(rng.integers(1, 4000))
        lv = np.minimum(rng.geometric(0.6, size=n) - 1, 7).astype(np.uint8)
        existing = int(rng.choice([0, 0, 1, 17, 5000, 200_000]))
        cur_max = -1 if existing == 0 else int(rng.integers(0, 5))
        max_batch, growth_div = int(rng.choice([1, 7, 256, 16384])), int(rng.choice([1, 4, 32]))
        got = np.zeros(n, dtype=np.uint64)
        ng = hl.hl_schedule(C.c_uint64(existing), cur_max, lv.ctypes.data_as(C.c_void_p), C.c_uint64(n), C.c_uint64(max_batch),
                            C.c_uint64(growth_div), C.c_int64(-1), got.ctypes.data_as(C.c_void_p))
        sizes = got[:ng].astype(np.int64)
        assert sizes.sum() == n and sizes.min() >= 1
        start, nodes, top = 0, existing, cur_max
        for b in sizes:
            rows = lv[start:start + b]
            if nodes == 0:
                assert b == 1
            else:
                assert b <= max(1, min(max_batch, nodes // growth_div))
            if b > 1:
                assert int(rows.max()) <= top          # nobody in a shared batch rises above the entry's level
            elif int(rows[0]) <= top and nodes and start + 1 < n and max(1, min(max_batch, nodes // growth_div)) > 1:
                assert int(lv[start + 1]) > top        # a singleton that is not a promotion: the next row is one
            top = max(top, int(rows.max()))
            start, nodes = start + int(b), nodes + int(b)


def test_search_shape_invariants_over_the_option_space(hl):
    """Whatever the options: a team only ever runs inside the solo shape, with a compute unit per query and a variant that
    exists; touches only inside the solo shape and only up to one query per compute unit; RowTouch never for rows wider than
    the window of whoever does the touching."""
    rng = np.random.default_rng(7)
    for trial in range(400):
        n = int(rng.choice([1, 2, 31, 32, 33, 204, 255, 256, 257, 1024, 5000]))
        dim = int(rng.choice([3, 16, 96, 100, 128, 256, 384, 512, 768, 1536]))
        M0 = int(rng.choice([4, 32, 64, 96]))
        kw = dict(solo_mode=int(rng.integers(0, 3)), team=bool(rng.integers(0, 2)), touch_rows=bool(rng.integers(0, 2)),
                  touch_lists=bool(rng.integers(0, 2)), n_cus=int(rng.choice([64, 256])), force_looping=bool(rng.integers(0, 2)),
                  solo_lds=int(rng.choice([20_000, 70_000, 163_000])), crew=bool(rng.integers(0, 2)),
                  engine_walkers=int(rng.choice([0, 0, 1, 4])))
        solo, team, touch, crew, roomy = _shape(hl, n, dim, M0, full=True, **kw)
        V = (dim + 3) // 4
        lines = (V * 16 + 127) // 128
        if kw["solo_mode"] == 0:
            assert not solo
        if kw["solo_mode"] == 2:
            assert solo
        if not solo:
            assert not team and (touch & 0xFF) == 0 and crew == kw["crew"]
            if touch:  # ListTouch in the workgroup engine: crews only, one walker per workgroup, a compute unit per query
                assert crew and roomy and n <= kw["n_cus"]
            assert roomy == (kw["engine_walkers"] == 1 if kw["engine_walkers"] else n <= kw["n_cus"])
        else:
            assert not crew and roomy
        if team:
            assert kw["team"] and n <= kw["n_cus"] and kw["solo_lds"] + 528 <= 160 * 1024
        if touch:
            assert n <= 256
        if touch & 0xFF:
            assert kw["touch_rows"] and (touch & 0xFF) == lines and lines <= (8 if team else 4)
        if touch & 0x100:
            assert kw["touch_lists"]


def test_visited_set_sizing_keeps_mid_sized_searches_in_lds(hl):
    """Round 4: a walker's visited set stays in LDS up to 2^13 cells (32 KiB; four walkers per workgroup).  With
    64 cells per entry of the limit it left LDS from limit 129 on and every probe round became an L2 / memory round trip; the
    sizing rule now halves the table for limits 129-256 (queries that outgrow it are re-run), leaves limits up to 128 and beyond
    256 as they were, and never touches the build's tables."""
    hl.hl_search_visited_log2.restype = C.c_uint32
    hl.hl_build_visited_log2.restype = C.c_uint32

    def search(limit, bump=0, M0=64, cap_max=64, max_log2=24):
        return hl.hl_search_visited_log2(C.c_uint64(limit), C.c_uint32(bump), C.c_uint64(M0), C.c_uint64(cap_max), C.c_uint32(max_log2))

    def build(limit, bump=0, M0=64, cap_max=64, max_log2=24):
        return hl.hl_build_visited_log2(C.c_uint64(limit), C.c_uint32(bump), C.c_uint64(M0), C.c_uint64(cap_max), C.c_uint32(max_log2))

    LDS_MAX = 13
    for limit in (10, 60, 64, 100, 128):                       # the headline (ef 60) and everything up to 128: as before, in LDS
        assert search(limit) == build(limit) == 13
    for limit in (129, 160, 192, 256):                         # halved: still 2^13, in LDS (64 cells per entry gave 2^14: HBM)
        assert search(limit) == LDS_MAX and build(limit) == 14
    for limit in (257, 384, 480, 512, 2000):                   # beyond: the roomy table, in HBM either way
        assert search(limit) == build(limit) > LDS_MAX
    assert search(100_000_000) == search(1 << 20)              # the limit is clamped (a table is never sized past 2^26 cells)
    # a retry is larger; nothing grows past "every node fits"
    for limit in (60, 200, 480):
        assert search(limit, bump=2) == search(limit) + 2
        assert search(limit, bump=2, max_log2=12) == 12
    # small M0 / narrow lists: never below 1024 cells, never below 8 x the widest list
    assert search(1, M0=4, cap_max=64) == 10 and search(1, M0=4, cap_max=512) == 12
    # monotone in the limit within each regime
    vals = [search(limit) for limit in range(1, 129)]
    assert vals == sorted(vals)


def test_compact_visited_set_is_an_exact_set(hl):
    """Round 4: the 16-bit form of the visited set — tag + displacement, csrc/visited_compact.h, the arithmetic the
    device code runs — must behave as a SET of slots: the map slot -> (home cell, tag) is one-to-one over all 2^24 slots, a
    sequential model of the table answers membership exactly like a Python set for every key it could place (whatever the
    load, with repeats, with runs of neighbouring slots), and the only other answer is "does not fit" (the device reports an
    overflow and the host re-runs the query with the 32-bit table) — never a wrong "seen before"."""
    hl.hl_compact_model.restype = C.c_uint64
    rng = np.random.default_rng(11)
    for L in (11, 14, 15):                                     # the forced-small table of the GPU probe; four walkers; one walker
        keys = np.arange(1 << 24, dtype=np.uint32)
        cells, tags = np.empty_like(keys), np.empty_like(keys)
        hl.hl_compact_home(keys.ctypes.data_as(C.c_void_p), C.c_uint64(len(keys)), C.c_uint32(L), cells.ctypes.data_as(C.c_void_p),
                           tags.ctypes.data_as(C.c_void_p))
        assert cells.max() == (1 << L) - 1 and tags.max() == (1 << (24 - L)) - 1
        assert len(np.unique((cells.astype(np.uint64) << 32) | tags)) == 1 << 24     # one-to-one
        assert np.bincount(cells, minlength=1 << L).max() == 1 << (24 - L)            # and perfectly even over the cells
    # ... and the 25-bit form (indexes of 2^24 .. 2^25 slots per GPU): one-to-one over all 2^25 slots
    form25 = 14 | (25 << 8)
    keys = np.arange(1 << 25, dtype=np.uint32)
    cells, tags = np.empty_like(keys), np.empty_like(keys)
    hl.hl_compact_home(keys.ctypes.data_as(C.c_void_p), C.c_uint64(len(keys)), C.c_uint32(form25), cells.ctypes.data_as(C.c_void_p),
                       tags.ctypes.data_as(C.c_void_p))
    assert cells.max() == (1 << 14) - 1 and tags.max() == (1 << 11) - 1
    assert len(np.unique((cells.astype(np.uint64) << 32) | tags)) == 1 << 25
    del keys, cells, tags
    for L, n_keys, universe in ((14, 3000, 12_500_000), (14, 12_000, 1 << 24), (15, 20_000, 10_000_000), (11, 1500, 40_000),
                                (14, 16_000, 20_000), (9, 400, 1 << 24), (form25, 3000, 30_000_000), (form25, 8000, 1 << 25)):
        base = rng.integers(0, universe, n_keys, dtype=np.uint32)
        runs = (base[: n_keys // 4, None] + np.arange(8, dtype=np.uint32)[None, :]).reshape(-1) % np.uint32(universe)
        keys = np.concatenate([base, runs, rng.permutation(base)]).astype(np.uint32)   # repeats: every key at least twice
        out = np.zeros(len(keys), dtype=np.uint8)
        used = hl.hl_compact_model(keys.ctypes.data_as(C.c_void_p), C.c_uint64(len(keys)), C.c_uint32(L), out.ctypes.data_as(C.c_void_p))
        seen, placed = set(), 0
        for key, got in zip(keys.tolist(), out.tolist()):
            if got == 2:                                       # could not be placed: allowed only for a key that is NOT in the table
                assert key not in seen
                continue
            assert got == (1 if key in seen else 0), (L, key, got)
            if got == 0:
                seen.add(key)
                placed += 1
        assert used == placed == len(seen) <= 1 << (L & 0xFF)
        # at the loads the engine allows (3/4 of the cells) with 6 displacement bits or more, a key that does not fit is rare
        if 14 <= L < 256 and len(set(keys.tolist())) <= (3 << L) // 8:
            assert (out == 2).mean() < 1e-3, (L, (out == 2).sum())


def test_compact_visited_set_policy(hl):
    """Which launches take it: the workgroup engine, limits of the 8-register list (257-512) whose 32-bit table would leave
    LDS, every slot within 24 bits, first pass only — and its cells are twice the words of the LDS table it lies over."""
    hl.hl_compact_cells_log2.restype = C.c_uint32

    def cells(plain_fits=False, solo=False, reg_list=True, limit=480, nodes=12_500_000, first=True, lds_log2=13):
        return hl.hl_compact_cells_log2(int(plain_fits), int(solo), int(reg_list), C.c_uint64(limit), C.c_uint64(nodes), int(first),
                                        C.c_uint32(lds_log2))

    assert cells() == 14 and cells(lds_log2=14) == 15 and cells(lds_log2=10) == 11   # four walkers / one walker / the forced-small table
    assert cells(limit=257) == 14 and cells(limit=512) == 14
    assert cells(limit=256) == 0 and cells(limit=60) == 0      # smaller limits: other instantiations (their tables fit LDS anyway)
    assert cells(reg_list=False) == 0                          # limits beyond 512: the list lives in HBM, E = 0
    assert cells(plain_fits=True) == 0                         # a small index: the plain table fits
    assert cells(solo=True) == 0                               # solo / team shapes
    assert cells(first=False) == 0                             # the re-run of an overflowing query: the plain table, larger
    # slots beyond 24 bits: the 25-bit form (round 6: one tag bit more, one displacement bit less); beyond 25 bits the plain set
    assert cells(nodes=1 << 24) == 14 and cells(nodes=(1 << 24) + 1) == (14 | (25 << 8)) and cells(nodes=1 << 25) == (14 | (25 << 8))
    assert cells(nodes=(1 << 25) + 1) == 0
    assert cells(lds_log2=7) == 0 and cells(lds_log2=16) == 0  # no displacement bit / fewer than eight tag bits


def test_exact_tile_lds_swizzle_is_a_conflict_free_bijection():
    """The LDS image of k_exact_scores_v4 (csrc/exact_kernels.h), restated: an LDS-DMA instruction lands its 64 lanes'
    16 bytes in 1 KiB of contiguous LDS — 8 rows x 8 positions of 16 bytes, 128 bytes per row, no padding — so bank conflicts are
    avoided by an XOR swizzle applied on BOTH sides: the lane that fills position p of row r fetches logical chunk p ^ (r & 7), the
    MFMA operand read of logical chunk c of row r goes to position c ^ (r & 7).  Checked here: the two maps are inverse to each
    other (every chunk of a row is written exactly once and read where it was written), a row's eight lanes still fetch its
    whole 128-byte line, and the operand reads — lanes 0-31 rows 0-31, lanes 32-63 the same rows one chunk further — touch 32
    distinct 16-byte bank groups per 8 lanes (64 banks of 4 bytes: no two lanes of an 8-lane pass share a bank)."""
    for piece_row0 in range(0, 128, 8):                      # one DMA instruction = 8 consecutive rows of the image
        written = {}
        for lane in range(64):
            r, p = piece_row0 + (lane >> 3), lane & 7
            chunk = p ^ (lane >> 3)                           # what the kernel fetches: (l & 7) ^ (l >> 3), row & 7 == l >> 3
            assert chunk == p ^ (r & 7)
            written[(r, p)] = chunk
            assert (piece_row0 * 128 + lane * 16) == r * 128 + p * 16   # lane-linear destination == [row][position]
        for r in range(piece_row0, piece_row0 + 8):
            assert sorted(written[(r, p)] for p in range(8)) == list(range(8))   # the row's full 128-byte line, each chunk once
            for c in range(8):
                assert written[(r, c ^ (r & 7))] == c                             # read side finds chunk c where it was put
    for g in range(4):                                        # k-group g of a step: half h of the wave owns chunk 2 g + h
        for first in range(0, 64, 8):                         # an 8-lane pass of ds_read_b128
            banks = set()
            for lane in range(first, first + 8):
                row, chunk = lane & 31, 2 * g + (lane >> 5)
                byte = row * 128 + ((chunk ^ (lane & 7)) * 16)
                assert (chunk ^ (lane & 7)) == (chunk ^ (row & 7))
                for w in range(4):
                    banks.add((byte // 4 + w) % 64)
            assert len(banks) == 32                            # 8 lanes x 4 dwords, all in different banks
