"""ctypes driver for the two CPU checkers under oracle/ (TEST INFRASTRUCTURE ONLY).

`load_oracle()`  -> oracle/liboracle.so            the repo's restatement
`load_ref()`     -> oracle/_ref/libusearch_ref.so  the reference's vendored usearch (None if not built)

Both export the surface declared in oracle/oracle_api.h; `CpuIndex` wraps one handle of either.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

METRICS = {"l2sq": 0, "cosine": 1, "ip": 2}

_u64, _i64, _vp, _int = C.c_uint64, C.c_int64, C.c_void_p, C.c_int


def _declare(lib, oracle_ext):
    lib.orc_create.restype = _vp
    lib.orc_create.argtypes = [_u64, _int, _u64, _u64, _u64, _u64]
    lib.orc_destroy.argtypes = [_vp]
    lib.orc_last_error.restype = C.c_char_p
    lib.orc_last_error.argtypes = [_vp]
    lib.orc_reserve.argtypes = [_vp, _u64, _u64]
    lib.orc_add.argtypes = [_vp, _i64, _vp, _vp]
    lib.orc_search.restype = _u64
    lib.orc_search.argtypes = [_vp, _vp, _u64, _u64, _int, _vp, _vp, _vp]
    lib.orc_search_filtered.restype = _u64
    lib.orc_search_filtered.argtypes = [_vp, _vp, _u64, _u64, _vp, _u64, _vp, _vp, _vp]
    lib.orc_remove.restype = _u64
    lib.orc_remove.argtypes = [_vp, _i64]
    lib.orc_compact.argtypes = [_vp]
    for name in ("orc_size", "orc_nodes", "orc_capacity", "orc_max_level", "orc_serialized_length"):
        getattr(lib, name).restype = _u64
        getattr(lib, name).argtypes = [_vp]
    lib.orc_level_stats.argtypes = [_vp, _u64, _vp]
    lib.orc_save.restype = _i64
    lib.orc_save.argtypes = [_vp, _vp, _u64]
    lib.orc_load.argtypes = [_vp, _vp, _u64]
    lib.orc_distance.restype = C.c_float
    lib.orc_distance.argtypes = [_int, _vp, _vp, _u64]
    if oracle_ext:
        lib.orc_set_mode.argtypes = [_vp, _int, _int]
        lib.orc_set_register_queue.argtypes = [_vp, _u64]
        lib.orc_register_queue_state.restype = _u64
        lib.orc_register_queue_state.argtypes = [_vp, _vp]
        lib.orc_set_pipeline_check.argtypes = [_vp, _int]
        lib.orc_pipeline_check_state.argtypes = [_vp, _vp]
        lib.orc_compact_dropping.argtypes = [_vp]
        lib.orc_compact_reordering.argtypes = [_vp]
        lib.orc_distance_wave.restype = C.c_float
        lib.orc_distance_wave.argtypes = [_int, _vp, _vp, _u64]
        lib.orc_draw_levels.argtypes = [_u64, _u64, _vp]
        lib.orc_schedule.restype = _u64
        lib.orc_schedule.argtypes = [_u64, _int, _vp, _u64, _u64, _u64, _vp]
        lib.orc_build_batch.argtypes = [_vp, _vp, _vp, _u64, _u64, _u64]
        lib.orc_node_level.argtypes = [_vp, _u64]
        lib.orc_node_key.restype = _i64
        lib.orc_node_key.argtypes = [_vp, _u64]
        lib.orc_neighbors.restype = _u64
        lib.orc_neighbors.argtypes = [_vp, _u64, _int, _vp]
        lib.orc_entry_slot.restype = _u64
        lib.orc_entry_slot.argtypes = [_vp]
        lib.orc_counters.argtypes = [_vp, _vp]
        lib.orc_array_function.restype = None
        lib.orc_array_function.argtypes = [_int, _vp, _vp, _int, _u64, _u64, _vp]
    else:
        lib.orc_search_mt.restype = C.c_double
        lib.orc_search_mt.argtypes = [_vp, _vp, _u64, _u64, _u64, _u64, _u64, C.c_double, _vp, _vp]
        lib.orc_add_mt.restype = C.c_double
        lib.orc_add_mt.argtypes = [_vp, _vp, _vp, _u64, _u64, C.c_double, _vp]
    lib._oracle_ext = oracle_ext
    return lib


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])


def load_oracle():
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "hnsw_oracle.cpp")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build_oracle()
    return _declare(C.CDLL(path), True)


def load_ref():
    path = os.path.join(ORACLE_DIR, "_ref", "libusearch_ref.so")
    if not os.path.exists(path):
        if os.path.isdir("/root/reference/src/include/usearch"):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])
        if not os.path.exists(path):
            return None
    return _declare(C.CDLL(path), False)


def _p(a):
    return a.ctypes.data if a is not None else None


class CpuIndex:
    """One index handle of either CPU library (same call shapes as the reference's HNSWIndex uses)."""

    def __init__(self, lib, dim, metric="l2sq", M=16, M0=None, ef_construction=128, ef_search=64, order=0, wave=0):
        self.lib, self.dim, self.metric = lib, dim, metric
        self.M, self.M0 = M, (2 * M if M0 is None else M0)
        self.h = lib.orc_create(dim, METRICS[metric], self.M, self.M0, ef_construction, ef_search)
        self.ef_search = ef_search
        if order or wave:
            lib.orc_set_mode(self.h, order, wave)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_destroy(self.h)
            self.h = None

    def error(self):
        return self.lib.orc_last_error(self.h).decode()

    def reserve(self, members, threads=1):
        assert self.lib.orc_reserve(self.h, members, threads) == 0

    def add(self, key, vec):
        vec = np.ascontiguousarray(vec, dtype=np.float32)
        st = np.zeros(3, dtype=np.uint64)
        rc = self.lib.orc_add(self.h, int(key), _p(vec), _p(st))
        if rc:
            raise RuntimeError(self.error())
        return st

    def add_many(self, keys, vecs):
        vecs = np.ascontiguousarray(vecs, dtype=np.float32)
        stats = np.zeros((len(keys), 3), dtype=np.uint64)
        for i, k in enumerate(keys):
            rc = self.lib.orc_add(self.h, int(k), vecs[i].ctypes.data, stats[i].ctypes.data)
            if rc:
                raise RuntimeError(self.error())
        return stats

    def build_batch(self, keys, vecs, max_batch, growth_div):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        vecs = np.ascontiguousarray(vecs, dtype=np.float32)
        rc = self.lib.orc_build_batch(self.h, _p(keys), _p(vecs), len(keys), max_batch, growth_div)
        if rc:
            raise RuntimeError(self.error())

    def search(self, q, k, ef=None, exact=False):
        q = np.ascontiguousarray(q, dtype=np.float32)
        keys = np.zeros(k, dtype=np.int64)
        d = np.zeros(k, dtype=np.float32)
        st = np.zeros(2, dtype=np.uint64)
        n = self.lib.orc_search(self.h, _p(q), k, self.ef_search if ef is None else ef, int(exact), _p(keys), _p(d), _p(st))
        return keys[:n], d[:n], st

    def search_many(self, Q, k, ef=None, exact=False):
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        keys = np.full((len(Q), k), -1, dtype=np.int64)
        d = np.full((len(Q), k), np.inf, dtype=np.float32)
        st = np.zeros((len(Q), 2), dtype=np.uint64)
        cnt = np.zeros(len(Q), dtype=np.int64)
        e = self.ef_search if ef is None else ef
        for i in range(len(Q)):
            cnt[i] = self.lib.orc_search(self.h, Q[i].ctypes.data, k, e, int(exact), keys[i].ctypes.data,
                                         d[i].ctypes.data, st[i].ctypes.data)
        return keys, d, cnt, st

    def search_many_filtered(self, Q, k, ef, allowed_bitmap, n_bits):
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        allowed_bitmap = np.ascontiguousarray(allowed_bitmap, dtype=np.uint64)
        keys = np.full((len(Q), k), -1, dtype=np.int64)
        d = np.full((len(Q), k), np.inf, dtype=np.float32)
        st = np.zeros((len(Q), 2), dtype=np.uint64)
        cnt = np.zeros(len(Q), dtype=np.int64)
        for i in range(len(Q)):
            cnt[i] = self.lib.orc_search_filtered(self.h, Q[i].ctypes.data, k, ef, _p(allowed_bitmap), n_bits,
                                                  keys[i].ctypes.data, d[i].ctypes.data, st[i].ctypes.data)
        return keys, d, cnt, st

    def set_register_queue(self, cap):
        """Model the engine's register queue of `cap` pending candidates in tombstone / predicate searches (0 = off)."""
        self.lib.orc_set_register_queue(self.h, int(cap))

    def set_pipeline_check(self, on=True):
        """Kernel mode only: every plain expansion predicts its successor by the engine's pipelined rule and compares."""
        self.lib.orc_set_pipeline_check(self.h, int(bool(on)))

    def pipeline_check_state(self):
        """(expansions checked, left to the plain order because of an exact tie / NaN, WRONG predictions)."""
        out = np.zeros(3, dtype=np.uint64)
        self.lib.orc_pipeline_check_state(self.h, _p(out))
        return int(out[0]), int(out[1]), int(out[2])

    def register_queue_state(self):
        """(overflowed, harmless drops) since the last call."""
        drops = C.c_uint64(0)
        over = self.lib.orc_register_queue_state(self.h, C.byref(drops))
        return bool(over), int(drops.value)

    def remove(self, key):
        return self.lib.orc_remove(self.h, int(key))

    def compact(self):
        assert self.lib.orc_compact(self.h) == 0

    def compact_dropping(self):
        """The ENGINE's compaction (drops tombstones; DESIGN.md deviation Q3), mirrored by the oracle only."""
        assert self.lib.orc_compact_dropping(self.h) == 0

    def compact_reordering(self):
        """The ENGINE's vss_compact: the reference's (level, cluster) order + pruning, mirrored by the oracle only."""
        assert self.lib.orc_compact_reordering(self.h) == 0

    def size(self):
        return self.lib.orc_size(self.h)

    def nodes(self):
        return self.lib.orc_nodes(self.h)

    def capacity(self):
        return self.lib.orc_capacity(self.h)

    def max_level(self):
        return self.lib.orc_max_level(self.h)

    def level_stats(self, level):
        out = np.zeros(4, dtype=np.uint64)
        self.lib.orc_level_stats(self.h, level, _p(out))
        return out

    def save(self):
        n = self.lib.orc_serialized_length(self.h)
        buf = np.zeros(n, dtype=np.uint8)
        w = self.lib.orc_save(self.h, _p(buf), n)
        if w < 0:
            raise RuntimeError(self.error())
        return buf[:w].tobytes()

    def load(self, blob):
        buf = np.frombuffer(blob, dtype=np.uint8).copy()
        rc = self.lib.orc_load(self.h, _p(buf), len(buf))
        if rc:
            raise RuntimeError(self.error())

    def load_buffer(self, buf, length):
        """Load from a uint8 numpy buffer without copying it first."""
        rc = self.lib.orc_load(self.h, _p(buf), length)
        if rc:
            raise RuntimeError(self.error())

    # ---- reference build only: the reference's own multi-threaded entry points (bench.py cpu_baseline) ----
    def search_mt(self, Q, k, ef, threads, total_queries, max_seconds=1e9):
        """Returns (seconds, queries done, keys of the first len(Q) queries)."""
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        keys = np.full((len(Q), k), -1, dtype=np.int64)
        done = C.c_uint64(0)
        s = self.lib.orc_search_mt(self.h, _p(Q), len(Q), k, ef, threads, total_queries, max_seconds, _p(keys),
                                   C.addressof(done))
        if s < 0:
            raise RuntimeError("orc_search_mt failed")
        return s, done.value, keys

    def add_mt(self, keys, vecs, threads, max_seconds=1e9):
        """Returns (seconds, rows added)."""
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        vecs = np.ascontiguousarray(vecs, dtype=np.float32)
        done = C.c_uint64(0)
        s = self.lib.orc_add_mt(self.h, _p(keys), _p(vecs), len(keys), threads, max_seconds, C.addressof(done))
        if s < 0:
            raise RuntimeError("orc_add_mt failed")
        return s, done.value

    # ---- oracle-only ----
    def neighbors(self, slot, level):
        out = np.zeros(max(self.M, self.M0), dtype=np.uint32)
        n = self.lib.orc_neighbors(self.h, slot, level, _p(out))
        return out[:n].copy()

    def node_level(self, slot):
        return self.lib.orc_node_level(self.h, slot)

    def node_key(self, slot):
        return self.lib.orc_node_key(self.h, slot)

    def entry_slot(self):
        return self.lib.orc_entry_slot(self.h)


def parse_stream(blob):
    """Decode the reference's serialized stream into plain arrays."""
    b = np.frombuffer(blob, dtype=np.uint8)
    rows, bpv = np.frombuffer(b[:8].tobytes(), dtype=np.uint32)
    rows, bpv = int(rows), int(bpv)
    off = 8
    vectors = np.frombuffer(b[off:off + rows * bpv].tobytes(), dtype=np.float32).reshape(rows, bpv // 4) if rows else None
    off += rows * bpv
    head = b[off:off + 64].tobytes()
    off += 64
    gh = np.frombuffer(b[off:off + 40].tobytes(), dtype=np.uint64)
    off += 40
    size, M, M0, max_level, entry = (int(x) for x in gh)
    levels = np.frombuffer(b[off:off + 2 * size].tobytes(), dtype=np.int16)
    off += 2 * size
    keys = np.zeros(size, dtype=np.int64)
    adj = []
    for i in range(size):
        keys[i] = np.frombuffer(b[off:off + 8].tobytes(), dtype=np.int64)[0]
        lvl = int(np.frombuffer(b[off + 8:off + 10].tobytes(), dtype=np.int16)[0])
        assert lvl == levels[i]
        off += 10
        per_level = []
        for l in range(lvl + 1):
            cap = M0 if l == 0 else M
            rec = np.frombuffer(b[off:off + 4 + 4 * cap].tobytes(), dtype=np.uint32)
            per_level.append(rec[1:1 + int(rec[0])].copy())
            off += 4 + 4 * cap
        adj.append(per_level)
    assert off == len(b)
    return dict(rows=rows, dim=bpv // 4, vectors=vectors, head=head, M=M, M0=M0, max_level=np.int64(max_level).item(),
                entry=entry, levels=levels, keys=keys, adj=adj)
