"""-m gpu: the data flow behind the reference's own SQL tests (test/sql/hnsw/*.test), replayed at the index boundary —
what reaches `usearch` in those tests reaches `libvssgpu.so` here, through the same sequence of calls HNSWIndex makes
(create / Construct / Delete / InitializeScan / PersistToDisk + reload / Compact / GetStats).  The planner-level
assertions of those tests (EXPLAIN shows HNSW_INDEX_SCAN, Binder messages) live above the boundary and are out of scope;
the option strings are covered by tests/test_host_harness.py."""
import numpy as np
import pytest

import datagen
import gpu_common as gc

pytestmark = pytest.mark.gpu

GRID = datagen.readme_grid()  # range(1,10)^3, row id = position: the table every reference test builds


def grid_index(metric="l2sq", **kw):
    idx = gc.gpu_index(3, metric, **kw)
    idx.reserve(len(GRID))
    idx.add(np.arange(len(GRID)), GRID)
    return idx


def reopen(idx, metric="l2sq"):
    """CHECKPOINT + restart: PersistToDisk writes the stream, the new process loads it (hnsw_index.cpp:223-236, 532-554)."""
    blob = idx.save()
    back = gc.gpu_index(3, metric)
    back.load(blob)
    return back


def array_distance(a, b):
    return np.sqrt(((np.asarray(a, dtype=np.float32) - np.asarray(b, dtype=np.float32)) ** 2).sum(-1))


def test_hnsw_basic_results_survive_checkpoint_and_restart():
    """hnsw_basic.test:27-55 — the three nearest rows of [1,2,3] are closer than 1.5, before and after a restart."""
    idx = grid_index()
    q = np.array([1, 2, 3], dtype=np.float32)
    for index in (idx, reopen(idx)):
        rows = index.search(q, 3)
        assert len(rows) == 3 and np.all(array_distance(GRID[rows], q) < 1.5)


def test_hnsw_crud_insert_delete_restart_compact():
    """hnsw_crud.test:16-50 — one row indexed, checkpoint, one more inserted, the first deleted, restart: the scan for
    [1,2,3] returns only [5,5,5]; PRAGMA hnsw_compact_index then succeeds."""
    idx = gc.gpu_index(3, "l2sq")
    idx.reserve(32)
    idx.add(np.array([0]), np.array([[1.0, 2.0, 3.0]], dtype=np.float32))
    idx = reopen(idx)                      # CHECKPOINT (the index keeps running; a reload is the strictest form of it)
    idx.reserve(32)
    idx.add(np.array([1]), np.array([[5.0, 5.0, 5.0]], dtype=np.float32))
    assert idx.remove(np.array([0])) == 1
    idx = reopen(idx)                      # restart
    rows = idx.search(np.array([1, 2, 3], dtype=np.float32), 3)
    assert rows.tolist() == [1]
    idx.compact()
    assert idx.size() == 1 and idx.search(np.array([1, 2, 3], dtype=np.float32), 3).tolist() == [1]


def test_hnsw_insert_counts_across_restarts():
    """hnsw_insert.test / hnsw_insert_wal.test — pragma_hnsw_index_info().count is 0 for the empty index, 2 after two
    inserts following a restart, still 2 after another restart, 3 after one more insert, 3 after the last restart."""
    idx = gc.gpu_index(3, "l2sq")
    assert idx.size() == 0
    idx = reopen(idx)
    assert idx.size() == 0
    idx.reserve(32)                        # HNSWIndex::Construct grows to NextPowerOfTwo (hnsw_index.cpp:443-461)
    idx.add(np.array([0]), np.array([[1.0, 2.0, 3.0]], dtype=np.float32))
    idx.add(np.array([1]), np.array([[4.0, 5.0, 6.0]], dtype=np.float32))
    assert idx.size() == 2
    idx = reopen(idx)
    assert idx.size() == 2
    idx.reserve(32)
    idx.add(np.array([2]), np.array([[7.0, 8.0, 9.0]], dtype=np.float32))
    assert idx.size() == 3
    idx = reopen(idx)
    assert idx.size() == 3
    assert sorted(idx.search(np.array([4, 5, 6], dtype=np.float32), 3).tolist()) == [0, 1, 2]


@pytest.mark.parametrize("metric,fn", [("l2sq", "array_distance"), ("cosine", "array_cosine_distance"),
                                       ("ip", "array_negative_inner_product")])
def test_hnsw_metrics_each_index_ranks_like_its_sql_function(metric, fn):
    """hnsw_metrics.test — one index per metric on the same table; ORDER BY <function>(vec, [1,2,3]) LIMIT 3 served by
    the index of the matching metric returns what the function itself ranks first (ties allowed: the grid is full of
    them, so the returned rows must carry the three smallest function VALUES)."""
    idx = grid_index(metric)
    q = np.array([1, 2, 3], dtype=np.float32)
    rows = idx.search(q, 3, ef=128)
    values = gc.pkg().distance_batch(fn, GRID, q)          # the SQL function over the whole column
    assert len(rows) == 3
    assert np.allclose(np.sort(values[rows]), np.sort(values)[:3], rtol=1e-6, atol=1e-7)


def test_hnsw_topk_min_by_rewrite():
    """hnsw_topk.test:27-31 — min_by(vec, array_distance(vec, [5,5,5]), 3): the coordinates of the three rows sum to a
    value between 45 and 50."""
    idx = grid_index()
    rows = idx.search(np.array([5, 5, 5], dtype=np.float32), 3)
    assert 45 <= float(GRID[rows].sum()) <= 50


def test_where_clause_rows_are_the_nearest():
    """where_clause_segfault.test:23-32 — nine copies of the grid (id 1..9), top-3 of [1,2,3] all at distance < 1.0."""
    vecs = np.tile(GRID, (9, 1))
    idx = gc.gpu_index(3, "l2sq")
    idx.reserve(len(vecs))
    idx.add(np.arange(len(vecs)), vecs)
    rows = idx.search(np.array([1, 2, 3], dtype=np.float32), 3)
    assert len(rows) == 3 and np.all(array_distance(vecs[rows], [1, 2, 3]) < 1.0)


def test_lateral_join_two_row_tables():
    """hnsw_lateral_join.test:13-52 — b = {[4,5,6], [1,2,3]} indexed, a = {[1,2,3], [4,5,6]} probes it as ONE batch:
    LIMIT 1 joins every a row to its identical b row at distance 0.0; LIMIT 2 returns both b rows, nearest first, with
    or without a NULL vector appended to b (NULLs never reach the index: hnsw_index.cpp:467-470)."""
    b = np.array([[4, 5, 6], [1, 2, 3]], dtype=np.float32)
    a = np.array([[1, 2, 3], [4, 5, 6]], dtype=np.float32)
    idx = gc.gpu_index(3, "l2sq")
    idx.reserve(8)
    idx.add(np.array([0, 1]), b)
    keys, dist, counts = idx.search_batch(a, 1)
    assert keys[:, 0].tolist() == [1, 0] and dist[:, 0].tolist() == [0.0, 0.0] and counts.tolist() == [1, 1]
    keys2, dist2, counts2 = idx.search_batch(a, 2)
    assert keys2.tolist() == [[1, 0], [0, 1]] and counts2.tolist() == [2, 2]
    assert dist2[:, 0].tolist() == [0.0, 0.0] and dist2[:, 1].tolist() == [27.0, 27.0]      # l2sq of (3,3,3)
    null_row = np.array([[np.nan, np.nan, np.nan]], dtype=np.float32)
    idx.add(np.array([2]), null_row, validity=np.array([0], dtype=np.uint64))               # INSERT (NULL, 'none')
    assert idx.size() == 2
    keys3, _, counts3 = idx.search_batch(a, 2)
    assert keys3.tolist() == [[1, 0], [0, 1]] and counts3.tolist() == [2, 2]


def _sql_array_distance(rows, q):
    """array_distance(vec, q) as the engine computes it (vss_distance_batch: SURVEY §8 row a13) — the value DuckDB's SELECT
    list / ORDER BY evaluates on the rows the index scan returned."""
    return gc.pkg().distance_batch("array_distance", np.ascontiguousarray(rows, dtype=np.float32), np.asarray(q, dtype=np.float32))


def test_array_function_expectations_of_the_reference_sql_tests():
    """every NUMERIC `array_*` expectation the reference's SQL tests hold, with the value computed by the
    engine's own array_distance kernel on the rows the engine's own index returned (a13 stays parity-unpinned against DuckDB
    v1.4.3 itself — its source is not in the reference tree — but these are the reference's committed results):
      hnsw_lateral_join.test:29-33   dist = 0.0 for both joined rows
      hnsw_basic.test:29-34, 50-55   array_distance([1,2,3], vec) < 1.5 -> true x 3, before and after the restart
      where_clause_segfault.test:23-32, 44-55, 74-95   array_distance(vec, [1,2,3]) < 1.0 -> 1 x 3, with WHERE id > 0 and
                                     WHERE id > 3 (here: pushed into the traversal), on both tables
      hnsw_result.test:23-28         0.0, 1.0, 1.0 (also tests/test_gpu_parity.py::test_array_distance_readme_values)"""
    q = np.array([1, 2, 3], dtype=np.float32)
    # hnsw_lateral_join.test
    b = np.array([[4, 5, 6], [1, 2, 3]], dtype=np.float32)
    a = np.array([[1, 2, 3], [4, 5, 6]], dtype=np.float32)
    idx = gc.gpu_index(3, "l2sq")
    idx.reserve(8)
    idx.add(np.array([0, 1]), b)
    keys, _, _ = idx.search_batch(a, 1)
    joined = b[keys[:, 0]]
    dist = gc.pkg().distance_batch("array_distance", a, joined)  # column operand: array_distance(a.a_vec, b.b_vec)
    assert dist.tolist() == [0.0, 0.0]
    # hnsw_basic.test
    idx = grid_index()
    for index in (idx, reopen(idx)):
        rows = index.search(q, 3)
        assert (_sql_array_distance(GRID[rows], q) < 1.5).tolist() == [True, True, True]
    # hnsw_result.test: the three values themselves, ascending as ORDER BY returns them
    rows = idx.search(q, 3)
    assert sorted(_sql_array_distance(GRID[rows], q).tolist()) == [0.0, 1.0, 1.0]
    # where_clause_segfault.test: ids x grid, two tables; WHERE id > t as a predicate over row ids
    for lo, hi in ((1, 10), (0, 10)):
        side = np.arange(lo, hi, dtype=np.float32)
        grid = np.stack(np.meshgrid(side, side, side, indexing="ij"), -1).reshape(-1, 3)
        ids = np.repeat(np.arange(lo, hi), len(grid))
        vecs = np.tile(grid, (hi - lo, 1)).astype(np.float32)
        index = gc.gpu_index(3, "l2sq")
        index.reserve(len(vecs))
        index.add(np.arange(len(vecs)), vecs)
        for threshold in (0, 3):
            allowed = np.zeros((len(vecs) + 63) // 64, dtype=np.uint64)
            for r in np.nonzero(ids > threshold)[0]:
                allowed[r >> 6] |= np.uint64(1) << np.uint64(r & 63)
            keys, _, cnt = index.search_batch_filtered(q[None, :], 3, 64, allowed, len(vecs))
            assert cnt[0] == 3 and np.all(ids[keys[0]] > threshold)
            assert (_sql_array_distance(vecs[keys[0]], q) < 1.0).astype(int).tolist() == [1, 1, 1]
