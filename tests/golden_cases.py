"""Cases shared by tests/golden/make_golden.py (run against the reference build, oracle/_ref) and
tests/test_oracle_golden.py (run against oracle/liboracle.so, compared with the committed vectors).

Each case is a pure function  lib -> {name: ndarray}  over the surface of oracle/oracle_api.h, so the very
same driver produces the golden vectors from the reference and the candidate vectors from the restatement.
"""
import numpy as np

import datagen
from oracle_lib import CpuIndex, parse_stream

# name, n, dim, metric, M, M0, efc, efs, k, normalize
BUILD_CASES = [
    ("grid_l2sq", 729, 3, "l2sq", 16, 32, 128, 64, 3, False),
    ("grid_cosine", 729, 3, "cosine", 16, 32, 128, 64, 3, False),
    ("grid_ip", 729, 3, "ip", 16, 32, 128, 64, 3, False),
    ("mix2k16_l2sq", 2000, 16, "l2sq", 16, 32, 128, 64, 10, False),
    ("mix2k16_cosine", 2000, 16, "cosine", 16, 32, 128, 64, 10, True),
    ("mix2k16_ip", 2000, 16, "ip", 16, 32, 128, 64, 10, True),
    ("mix2k16_l2sq_m3", 2000, 16, "l2sq", 3, 3, 100, 100, 10, False),
    ("mix1k7_l2sq_m4", 1000, 7, "l2sq", 4, 6, 20, 10, 5, False),
    ("mix3k128_cosine", 3000, 128, "cosine", 16, 32, 128, 64, 10, True),
    ("mix1k768_l2sq", 1000, 768, "l2sq", 16, 32, 128, 64, 10, False),
    # corners: M0 not 2M, odd dimension, ef_construction below the list capacities, k above ef_search, dim 1536
    ("mix2k5_ip_m6", 2000, 5, "ip", 6, 9, 30, 20, 7, True),
    ("mix1k100_cosine_efc8", 1000, 100, "cosine", 16, 32, 8, 8, 12, True),
    ("mix600x1536_ip", 600, 1536, "ip", 8, 16, 64, 32, 100, True),
]


def case_inputs(case):
    name, n, dim, metric, M, M0, efc, efs, k, normalize = case
    if name.startswith("grid"):
        X = datagen.readme_grid()
        Q = np.array([[1, 2, 3], [5, 5, 5], [9, 9, 9], [0.5, 3.25, 7.75], [4, 4, 4.5]], dtype=np.float32)
    else:
        seed = sum(ord(c) for c in name)
        X = datagen.mixture(n, dim, seed, normalize=normalize)
        Q = datagen.mixture(40, dim, seed + 1000, n_clusters=max(2, int(np.sqrt(n))), normalize=normalize)
    return X, Q


def _search_block(idx, Q, k, **kw):
    keys, d, cnt, st = idx.search_many(Q, k, **kw)
    return keys, d.view(np.uint32), cnt, st


def run_build_case(lib, case, **mode):
    name, n, dim, metric, M, M0, efc, efs, k, normalize = case
    X, Q = case_inputs(case)
    idx = CpuIndex(lib, dim, metric, M, M0, efc, efs, **mode)
    idx.reserve(len(X), 1)
    add_stats = idx.add_many(np.arange(len(X)), X)
    blob = idx.save()
    out = {
        "input_sha": np.frombuffer(bytes.fromhex(datagen.sha(X)), dtype=np.uint8),
        "stream_sha": np.frombuffer(bytes.fromhex(datagen.sha(blob)), dtype=np.uint8),
        "stream_len": np.array([len(blob)], dtype=np.int64),
        "add_stats": add_stats.astype(np.uint32),
        "shape": np.array([idx.size(), idx.capacity(), idx.max_level()], dtype=np.int64),
    }
    st = parse_stream(blob)
    out["levels"] = st["levels"].copy()
    out["degree0"] = np.array([len(a[0]) for a in st["adj"]], dtype=np.uint16)
    if name == "grid_l2sq":
        out["stream"] = np.frombuffer(blob, dtype=np.uint8).copy()
    for tag, kw in (("default", {}), ("ef16", {"ef": 16}), ("ef200", {"ef": 200}), ("exact", {"exact": True})):
        keys, dbits, cnt, sst = _search_block(idx, Q, k, **kw)
        out["s_%s_keys" % tag] = keys
        out["s_%s_dbits" % tag] = dbits
        out["s_%s_cnt" % tag] = cnt
        out["s_%s_stats" % tag] = sst.astype(np.uint32)
    return out


def run_crud_case(lib, **mode):
    """Incremental growth (power-of-two reserve, as HNSWIndex::Construct hnsw_index.cpp:443-461), deletes,
    slot reuse through update(), compact, save/load, deletes after load."""
    d = 12
    X = datagen.mixture(600, d, 4242)
    Q = datagen.mixture(20, d, 4243, n_clusters=24)
    idx = CpuIndex(lib, d, "l2sq", 8, 16, 40, 30, **mode)
    out = {}
    cap = 32
    idx.reserve(cap, 1)
    for i in range(300):
        if idx.nodes() + 1 > cap:
            cap *= 2
            idx.reserve(cap, 1)
        idx.add(i, X[i])
    out["stream_a"] = np.frombuffer(bytes.fromhex(datagen.sha(idx.save())), dtype=np.uint8)
    out["removed"] = np.array([idx.remove(k) for k in list(range(10, 60, 3)) + [10, 100000]], dtype=np.int64)
    out["shape_a"] = np.array([idx.size(), idx.nodes(), idx.capacity(), idx.max_level()], dtype=np.int64)
    for j, a in enumerate(_search_block(idx, Q, 5)):
        out["search_a%d" % j] = a
    stats = []
    for i in range(300, 330):
        if idx.nodes() + 1 > cap and idx.size() == idx.nodes():
            cap *= 2
            idx.reserve(cap, 1)
        stats.append(idx.add(i, X[i]))
    out["reuse_stats"] = np.array(stats).astype(np.int64)
    out["stream_b"] = np.frombuffer(bytes.fromhex(datagen.sha(idx.save())), dtype=np.uint8)
    for j, a in enumerate(_search_block(idx, Q, 5)):
        out["search_b%d" % j] = a
    idx.compact()
    out["stream_c"] = np.frombuffer(bytes.fromhex(datagen.sha(idx.save())), dtype=np.uint8)
    for j, a in enumerate(_search_block(idx, Q, 5, ef=50)):
        out["search_c%d" % j] = a
    for j, a in enumerate(_search_block(idx, Q, 7, exact=True)):
        out["search_x%d" % j] = a
    blob = idx.save()
    idx2 = CpuIndex(lib, d, "l2sq", 8, 16, 40, 30, **mode)
    idx2.load(blob)
    out["stream_d"] = np.frombuffer(bytes.fromhex(datagen.sha(idx2.save())), dtype=np.uint8)
    out["shape_d"] = np.array([idx2.size(), idx2.nodes(), idx2.capacity(), idx2.max_level()], dtype=np.int64)
    out["removed_after_load"] = np.array([idx2.remove(100)], dtype=np.int64)
    for j, a in enumerate(_search_block(idx2, Q, 5)):
        out["search_d%d" % j] = a
    idx2.reserve(1024, 1)
    out["add_after_load"] = np.array([idx2.add(i, X[i]) for i in range(400, 420)]).astype(np.int64)
    out["stream_e"] = np.frombuffer(bytes.fromhex(datagen.sha(idx2.save())), dtype=np.uint8)
    out["level_stats"] = np.array([idx2.level_stats(l) for l in range(3)]).astype(np.int64)
    return out


def run_reuse_case(lib, **mode):
    """Slot reuse through update() with a free ring that wraps (index_dense.hpp:1766-1793, index.hpp:1150-1277,
    2801-2859): which slot every re-inserted row lands in, every list, and the answers afterwards."""
    d = 12
    X = datagen.mixture(900, d, 4242)
    Q = datagen.mixture(30, d, 4243, n_clusters=24)
    idx = CpuIndex(lib, d, "l2sq", 8, 16, 40, 30, **mode)
    idx.reserve(1024, 1)
    out = {}
    idx.add_many(np.arange(300), X[:300])
    out["removed_a"] = np.array([idx.remove(k) for k in range(10, 130, 3)], dtype=np.int64)
    out["slots_a"] = idx.add_many(np.arange(300, 350), X[300:350])[:, 2].astype(np.int64)
    out["stream_a"] = np.frombuffer(bytes.fromhex(datagen.sha(idx.save())), dtype=np.uint8)
    out["removed_b"] = np.array([idx.remove(k) for k in range(140, 290, 2)], dtype=np.int64)  # 75 pushes: the ring wraps
    out["slots_b"] = idx.add_many(np.arange(350, 520), X[350:520])[:, 2].astype(np.int64)
    blob = idx.save()
    out["stream_b"] = np.frombuffer(bytes.fromhex(datagen.sha(blob)), dtype=np.uint8)
    out["keys_b"] = parse_stream(blob)["keys"].copy()
    out["shape_b"] = np.array([idx.size(), idx.nodes(), idx.capacity(), idx.max_level()], dtype=np.int64)
    for j, a in enumerate(_search_block(idx, Q, 5, ef=40)):
        out["search_b%d" % j] = a
    return out


def filter_bitmap(n_bits, seed, fraction):
    bits = datagen.uniforms(seed, n_bits) < fraction
    pad = (-n_bits) % 64
    return np.packbits(np.concatenate([bits, np.zeros(pad, dtype=bool)]).astype(np.uint8), bitorder="little").view(np.uint64)


def run_filtered_case(lib, **mode):
    """filtered_search (index_dense.hpp:625-629): predicate over row ids pushed into the traversal, with tombstones."""
    n, d = 2500, 16
    X = datagen.mixture(n, d, 606)
    Q = datagen.mixture(40, d, 607, n_clusters=50)
    idx = CpuIndex(lib, d, "l2sq", 16, 32, 128, 64, **mode)
    idx.reserve(n, 1)
    idx.add_many(np.arange(n) * 3, X)
    for key in range(0, 3 * n, 33):
        idx.remove(key)
    out = {}
    for tag, frac, k, ef in (("half", 0.5, 10, 64), ("rare", 0.02, 10, 40), ("most", 0.95, 5, 16)):
        bm = filter_bitmap(3 * n - 7, 700 + k, frac)  # keys >= n_bits are rejected
        keys, dd, cnt, st = idx.search_many_filtered(Q, k, ef, bm, 3 * n - 7)
        out["f_%s_keys" % tag], out["f_%s_dbits" % tag] = keys, dd.view(np.uint32)
        out["f_%s_cnt" % tag], out["f_%s_stats" % tag] = cnt, st.astype(np.uint32)
    return out


LEVEL_MS = (2, 3, 16, 32)


def run_levels_case(lib):
    """First 1000 draws of the level generator per connectivity, read back from a built index's stream."""
    out = {}
    pts = datagen.normals(77, (1000, 2)).astype(np.float32)
    for M in LEVEL_MS:
        idx = CpuIndex(lib, 2, "l2sq", M, M, 8, 8)
        idx.reserve(1000, 1)
        idx.add_many(np.arange(1000), pts)
        out["levels_M%d" % M] = parse_stream(idx.save())["levels"].copy()
    return out


DIST_DIMS = (3, 128, 768, 1536)


def distance_inputs(dim):
    A = datagen.normals(900 + dim, (250, dim)).astype(np.float32)
    B = datagen.normals(1900 + dim, (250, dim)).astype(np.float32)
    A[0] = 0            # exactly one zero norm
    A[1] = 0
    B[1] = 0            # both zero norms
    B[2] = A[2]         # equal vectors
    B[3] = -A[3]        # opposite
    A[4] *= 1e-20       # tiny
    B[5] *= 1e18        # huge
    return A, B


def run_distance_case(lib):
    import ctypes
    out = {}
    for dim in DIST_DIMS:
        A, B = distance_inputs(dim)
        for mi, m in enumerate(("l2sq", "cosine", "ip")):
            with np.errstate(over="ignore"):
                r = np.array([lib.orc_distance(mi, A[i].ctypes.data, B[i].ctypes.data, dim) for i in range(len(A))],
                             dtype=np.float32)
            out["dist_%s_%d" % (m, dim)] = r.view(np.uint32)
    return out


def run_all(lib, **mode):
    res = {}
    for case in BUILD_CASES:
        for k, v in run_build_case(lib, case, **mode).items():
            res["%s/%s" % (case[0], k)] = v
    for k, v in run_crud_case(lib, **mode).items():
        res["crud/%s" % k] = v
    for k, v in run_filtered_case(lib, **mode).items():
        res["filtered/%s" % k] = v
    for k, v in run_reuse_case(lib, **mode).items():
        res["reuse/%s" % k] = v
    for k, v in run_levels_case(lib).items():
        res["levels/%s" % k] = v
    for k, v in run_distance_case(lib).items():
        res["distance/%s" % k] = v
    return res
