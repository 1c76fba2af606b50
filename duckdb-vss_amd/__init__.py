"""duckdb-vss_amd — Python driver over libvssgpu.so (the MI355X-native engine behind the C ABI in include/vssgpu.h).

This module is plumbing for tests/ and bench.py: it loads the in-tree shared library with ctypes and exposes one
class, `GpuIndex`, whose methods map 1:1 onto the C entry points (which in turn map onto the calls the reference's
HNSWIndex makes on its usearch member — see include/vssgpu.h for the file:line of each).  There is NO fallback:
if the library is missing or no HIP device is present, construction raises.

The directory name carries a hyphen (it is the repo's package directory, not an importable dotted name); use
`__graft_entry__.load_package()` or importlib to import it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.environ.get("VSS_LIBRARY") or os.path.join(HERE, "libvssgpu.so")  # VSS_LIBRARY: debug builds only
CSRC = os.path.join(HERE, "csrc")

METRICS = {"l2sq": 0, "cosine": 1, "ip": 2}
FUNCTIONS = {"array_distance": 0, "array_cosine_distance": 1, "array_negative_inner_product": 2}
FREE_KEY = np.iinfo(np.int64).max

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-int-to-pointer-cast"]


TRANSLATION_UNITS = ["vss_engine.hip", "vss_exchange.hip", "kernels_l2sq.hip", "kernels_cosine.hip", "kernels_ip.hip"]


def build_library(force=False, extra_flags=(), out=None):
    """hipcc cross-compiles the engine for gfx950 (works without a GPU): four translation units in parallel, then a
    shared link.  Objects live in build/ (git-ignored)."""
    out = out or os.path.join(HERE, "libvssgpu.so")
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    srcs.append(os.path.join(ROOT, "include", "vssgpu.h"))
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return out
    objdir = os.path.join(HERE, "build", os.path.basename(out))
    os.makedirs(objdir, exist_ok=True)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags) + ["-I", os.path.join(ROOT, "include")]
    procs, objs = [], []
    for tu in TRANSLATION_UNITS:
        obj = os.path.join(objdir, tu.replace(".hip", ".o"))
        objs.append(obj)
        procs.append(subprocess.Popen(["hipcc"] + flags + ["-c", os.path.join(CSRC, tu), "-o", obj]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    return out


_u64, _i64, _vp, _int, _u32 = C.c_uint64, C.c_int64, C.c_void_p, C.c_int, C.c_uint32
WRITE_CB = C.CFUNCTYPE(_int, _vp, _vp, _u64)
READ_CB = C.CFUNCTYPE(_int, _vp, _vp, _u64)

# every symbol include/vssgpu.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "vss_version": (C.c_char_p, []),
    "vss_create": (_int, [_u64, _int, _u64, _u64, _u64, _u64, _int, C.POINTER(_vp)]),
    "vss_destroy": (None, [_vp]),
    "vss_last_error": (C.c_char_p, [_vp]),
    "vss_set_stream": (_int, [_vp, _vp]),
    "vss_synchronize": (_int, [_vp]),
    "vss_reserve": (_int, [_vp, _u64, _u64]),
    "vss_stage_batch": (_int, [_vp, _vp, _vp, _vp, _u64]),
    "vss_stage_batch_device": (_int, [_vp, _vp, _vp, _u64]),
    "vss_build_finalize": (_int, [_vp]),
    "vss_add_batch": (_int, [_vp, _vp, _vp, _vp, _u64]),
    "vss_set_build_params": (_int, [_vp, _u64, _u64]),
    "vss_set_build_reorder": (_int, [_vp, _int]),
    "vss_set_option": (_int, [_vp, C.c_char_p, _i64]),
    "vss_search": (_int, [_vp, _vp, _u64, _u64, _vp, _vp]),
    "vss_search_batch": (_int, [_vp, _vp, _u64, _u64, _u64, _vp, _vp, _vp]),
    "vss_search_batch_device": (_int, [_vp, _vp, _u64, _u64, _u64, _vp, _vp, _vp]),
    "vss_search_batch_filtered": (_int, [_vp, _vp, _u64, _u64, _u64, _vp, _u64, _vp, _vp, _vp]),
    "vss_search_batch_filtered_device": (_int, [_vp, _vp, _u64, _u64, _u64, _vp, _u64, _vp, _vp, _vp]),
    "vss_search_batch_device_begin": (_int, [_vp, _int, _vp, _u64, _u64, _u64, _vp, _vp, _vp]),
    "vss_search_multi_device_begin": (_int, [_vp, _int, _u64, _vp, _u64, _u64, _u64, _vp, _vp, _vp]),
    "vss_search_batch_end": (_int, [_vp, _int]),
    "vss_search_exact_batch": (_int, [_vp, _vp, _u64, _u64, _vp, _vp, _vp]),
    "vss_search_exact_batch_device": (_int, [_vp, _vp, _u64, _u64, _vp, _vp, _vp]),
    "vss_last_search_stats": (_int, [_vp, _vp]),
    "vss_last_search_query_stats": (_int, [_vp, _vp, _u64]),
    "vss_timing": (_int, [_vp, _vp, _int]),
    "vss_build_work": (_int, [_vp, _vp]),
    "vss_remove_batch": (_int, [_vp, _vp, _u64, _vp]),
    "vss_compact": (_int, [_vp]),
    "vss_compact_ex": (_int, [_vp, _int, _vp]),
    "vss_size": (_u64, [_vp]),
    "vss_nodes": (_u64, [_vp]),
    "vss_build_progress": (_int, [_vp, _vp, _vp]),
    "vss_capacity": (_u64, [_vp]),
    "vss_max_level": (_u64, [_vp]),
    "vss_memory_usage": (_u64, [_vp]),
    "vss_dimensions": (_u64, [_vp]),
    "vss_metric": (_int, [_vp]),
    "vss_level_stats": (_int, [_vp, _u64, _vp]),
    "vss_serialized_length": (_u64, [_vp]),
    "vss_save": (_int, [_vp, WRITE_CB, _vp]),
    "vss_load": (_int, [_vp, READ_CB, _vp]),
    "vss_distance_batch": (_int, [_int, _vp, _vp, _int, _u64, _u64, _vp, _int]),
    "vss_distance_batch_device": (_int, [_int, _vp, _vp, _int, _u64, _u64, _vp, _vp]),
    "vss_merge_topk_device": (_int, [_vp, _vp, _u64, _u64, _u64, _vp, _vp, _vp, _vp]),
    "vss_packed_block_bytes": (_u64, [_u64, _u64]),
    "vss_merge_topk_packed_device": (_int, [_vp, _u64, _u64, _u64, _vp, _vp, _vp, _vp]),
    "vss_exchange_available": (_int, []),
    "vss_exchange_last_error": (C.c_char_p, []),
    "vss_exchange_unique_id": (_int, [_vp]),
    "vss_exchange_init_rank": (_int, [C.POINTER(_vp), _int, _vp, _int, _int]),
    "vss_exchange_init_all": (_int, [C.POINTER(_vp), _int, C.POINTER(_int)]),
    "vss_exchange_adopt": (_int, [C.POINTER(_vp), _vp, _int, _int]),
    "vss_exchange_ranks": (_int, [_vp]),
    "vss_exchange_allgather": (_int, [_vp, _vp, _vp, _u64, _vp]),
    "vss_exchange_group_begin": (_int, []),
    "vss_exchange_group_end": (_int, []),
    "vss_exchange_destroy": (_int, [_vp]),
}

_lib = None


def load_library():
    """Load the in-tree libvssgpu.so (building it first if sources are newer). Raises if it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build_library()
        try:
            # torch ships its own libamdhip64.so.7; whichever HIP runtime is loaded first serves the whole process, and
            # torch cannot initialise on top of the system one.  Import it first so both share torch's runtime.
            import torch  # noqa: F401
        except Exception:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if hasattr(lib, "vss_debug_phase_ticks"):
            lib.vss_debug_phase_ticks.restype = _int
            lib.vss_debug_phase_ticks.argtypes = [_vp, _vp, _u64]
        _lib = lib
    return _lib


def _p(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a  # raw device pointer (int)


class VssError(RuntimeError):
    pass


class GpuIndex:
    """One HNSW index resident on one MI355X (same call shapes as the reference's HNSWIndex uses on usearch)."""

    def __init__(self, dim, metric="l2sq", M=16, M0=None, ef_construction=128, ef_search=64, device=0):
        self.lib = load_library()
        self.dim, self.metric = dim, metric
        self.M, self.M0 = M, (2 * M if M0 is None else M0)
        h = _vp()
        rc = self.lib.vss_create(dim, METRICS[metric], self.M, self.M0, ef_construction, ef_search, device, C.byref(h))
        if rc != 0 or not h.value:
            raise VssError("vss_create failed: no MI355X / HIP device %d available (the engine has no CPU fallback)" % device)
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.vss_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise VssError(self.lib.vss_last_error(self.h).decode())

    # ---- build
    def reserve(self, members, threads=1):
        self._check(self.lib.vss_reserve(self.h, members, threads))

    def set_build_params(self, max_batch, growth_div):
        self._check(self.lib.vss_set_build_params(self.h, max_batch, growth_div))

    def set_build_reorder(self, on=True):
        self._check(self.lib.vss_set_build_reorder(self.h, 1 if on else 0))

    def set_stream(self, stream_ptr):
        self._check(self.lib.vss_set_stream(self.h, stream_ptr))

    def stage(self, rowids, vecs, validity=None):
        rowids = np.ascontiguousarray(rowids, dtype=np.int64)
        vecs = np.ascontiguousarray(vecs, dtype=np.float32)
        if validity is not None:
            validity = np.ascontiguousarray(validity, dtype=np.uint64)
        self._check(self.lib.vss_stage_batch(self.h, _p(rowids), _p(vecs), _p(validity), len(rowids)))

    def stage_device(self, d_rowids, d_vecs, count):
        self._check(self.lib.vss_stage_batch_device(self.h, d_rowids, d_vecs, count))

    def build_finalize(self):
        self._check(self.lib.vss_build_finalize(self.h))

    def add(self, rowids, vecs, validity=None):
        rowids = np.ascontiguousarray(rowids, dtype=np.int64)
        vecs = np.ascontiguousarray(vecs, dtype=np.float32)
        if validity is not None:
            validity = np.ascontiguousarray(validity, dtype=np.uint64)
        self._check(self.lib.vss_add_batch(self.h, _p(rowids), _p(vecs), _p(validity), len(rowids)))

    # ---- search
    # ---- tuning (vss_set_option: results never depend on any of these; tests and A/B measurements only)
    def set_option(self, name, value):
        self._check(self.lib.vss_set_option(self.h, name.encode(), int(value)))

    def set_search_params(self, waves=16, walkers=0):
        self.set_option("search.walkers", 0)  # (so that any waves / walkers pair can be reached in two steps)
        self.set_option("search.waves", waves)
        self.set_option("search.walkers", walkers)

    def set_search_solo(self, mode=1, max_queries=0):
        self.set_option("search.solo", mode)
        if max_queries:
            self.set_option("search.solo_max_queries", max_queries)

    def set_search_team(self, on=True):
        self.set_option("search.team", int(bool(on)))

    def set_search_crew(self, on=True):
        self.set_option("search.crew", int(on))

    def set_search_pipelined(self, on=True):
        self.set_option("search.pipelined", int(bool(on)))

    def set_search_wide_lists(self, on=True):
        self.set_option("search.wide_lists", int(bool(on)))

    def set_search_visited_set(self, compact=True, lds_table_log2_max=0, cells_per_limit=0, retry_in_place=True):
        self.set_option("search.visited_compact", int(bool(compact)))
        self.set_option("search.visited_lds_log2_max", lds_table_log2_max)
        self.set_option("search.visited_cells_per_limit", cells_per_limit)
        self.set_option("search.retry_in_place", int(bool(retry_in_place)))

    def set_search_probe_wait(self, flag_wait=True):
        self.set_option("search.probe_flag_wait", int(bool(flag_wait)))

    def set_search_lookahead(self, max_active_walkers=2):
        self.set_option("search.lookahead", max_active_walkers)

    def search(self, q, k, ef=0):
        """vss_search: ONE query (HNSW_INDEX_SCAN).  Kept lean: this wrapper's own microseconds count against the call."""
        if not (type(q) is np.ndarray and q.dtype == np.float32 and q.flags.c_contiguous):
            q = np.ascontiguousarray(q, dtype=np.float32)
        out = np.empty(k, dtype=np.int64)  # the engine writes every cell (unused ones = -1)
        n = _u64(0)
        rc = self.lib.vss_search(self.h, q.ctypes.data, k, ef, out.ctypes.data, C.byref(n))
        if rc != 0:
            self._check(rc)
        return out[:n.value]

    def search_batch(self, Q, k, ef=0, exact=False):
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        nq = len(Q)
        keys = np.full((nq, k), -1, dtype=np.int64)
        d = np.full((nq, k), np.inf, dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        if exact:
            self._check(self.lib.vss_search_exact_batch(self.h, _p(Q), nq, k, _p(keys), _p(d), _p(cnt)))
        else:
            self._check(self.lib.vss_search_batch(self.h, _p(Q), nq, k, ef, _p(keys), _p(d), _p(cnt)))
        return keys, d, cnt

    def search_batch_filtered(self, Q, k, ef, allowed_bitmap, n_bits):
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        allowed_bitmap = np.ascontiguousarray(allowed_bitmap, dtype=np.uint64)
        nq = len(Q)
        keys = np.full((nq, k), -1, dtype=np.int64)
        d = np.full((nq, k), np.inf, dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        self._check(self.lib.vss_search_batch_filtered(self.h, _p(Q), nq, k, ef, _p(allowed_bitmap), n_bits, _p(keys), _p(d),
                                                       _p(cnt)))
        return keys, d, cnt

    def search_batch_device(self, d_Q, nq, k, ef, d_keys, d_dist, d_counts, exact=False):
        if exact:
            self._check(self.lib.vss_search_exact_batch_device(self.h, d_Q, nq, k, d_keys, d_dist, d_counts))
        else:
            self._check(self.lib.vss_search_batch_device(self.h, d_Q, nq, k, ef, d_keys, d_dist, d_counts))

    def search_begin(self, context, d_Q, nq, k, ef, d_keys, d_dist, d_counts):
        self._check(self.lib.vss_search_batch_device_begin(self.h, context, d_Q, nq, k, ef, d_keys, d_dist, d_counts))

    def search_multi_begin(self, context, d_Qs, per_batch, k, ef, d_keys, d_dists, d_counts):
        """Several batches (lists of device pointers, one entry per batch) answered by one launch; search_end completes."""
        n = len(d_Qs)
        tables = [(C.c_void_p * n)(*[int(p) if p else None for p in t]) for t in (d_Qs, d_keys, d_dists, d_counts)]
        self._check(self.lib.vss_search_multi_device_begin(self.h, context, n, tables[0], per_batch, k, ef, tables[1], tables[2],
                                                           tables[3]))

    def set_search_gating(self, on):
        self.set_option("search.gating", 1 if on else 0)

    def search_end(self, context):
        self._check(self.lib.vss_search_batch_end(self.h, context))

    def last_search_stats(self):
        out = np.zeros(4, dtype=np.uint64)
        self._check(self.lib.vss_last_search_stats(self.h, _p(out)))
        return out

    def timing(self, reset=False):
        out = np.zeros(6, dtype=np.float64)
        self._check(self.lib.vss_timing(self.h, _p(out), int(reset)))
        return dict(search_kernel_ms=out[0], build_phase_a_ms=out[1], build_phase_b_ms=out[2], build_wall_ms=out[3],
                    build_batches=int(out[4]), build_retries=int(out[5]))

    def build_work(self):
        out = np.zeros(3, dtype=np.uint64)
        self._check(self.lib.vss_build_work(self.h, _p(out)))
        return dict(insert_distances=int(out[0]), insert_expansions=int(out[1]), link_distances=int(out[2]))

    def last_query_stats(self, nq):
        out = np.zeros((nq, 2), dtype=np.uint32)
        self._check(self.lib.vss_last_search_query_stats(self.h, _p(out), nq))
        return out

    # ---- maintenance
    def remove(self, rowids):
        rowids = np.ascontiguousarray(rowids, dtype=np.int64)
        n = _u64(0)
        self._check(self.lib.vss_remove_batch(self.h, _p(rowids), len(rowids), C.byref(n)))
        return n.value

    def compact(self, reorder=True):
        """vss_compact: drop tombstones + the reference's (level, cluster) reordering; reorder=False only prunes.
        Returns True if the rows were reordered."""
        done = C.c_int(0)
        self._check(self.lib.vss_compact_ex(self.h, 1 if reorder else 0, C.byref(done)))
        return bool(done.value)

    def size(self):
        return self.lib.vss_size(self.h)

    def nodes(self):
        return self.lib.vss_nodes(self.h)

    def capacity(self):
        return self.lib.vss_capacity(self.h)

    def max_level(self):
        return self.lib.vss_max_level(self.h)

    def build_progress(self):
        """(rows linked so far, rows of that build) — callable from another thread while build_finalize / add runs."""
        a, b = _u64(0), _u64(0)
        self.lib.vss_build_progress(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def memory_usage(self):
        return self.lib.vss_memory_usage(self.h)

    def level_stats(self, level):
        out = np.zeros(4, dtype=np.uint64)
        self._check(self.lib.vss_level_stats(self.h, level, _p(out)))
        return out

    # ---- persistence
    def save(self):
        chunks = []

        def write(ctx, data, size):
            chunks.append(C.string_at(data, size))
            return 1
        cb = WRITE_CB(write)
        self._check(self.lib.vss_save(self.h, cb, None))
        return b"".join(chunks)

    def serialized_length(self):
        return self.lib.vss_serialized_length(self.h)

    def save_into(self, buf):
        """Serialise into a preallocated uint8 numpy buffer (no intermediate copies; for multi-GB indexes)."""
        state = {"off": 0}
        base = buf.ctypes.data

        def write(ctx, data, size):
            if state["off"] + size > buf.nbytes:
                return 0
            C.memmove(base + state["off"], data, size)
            state["off"] += size
            return 1
        cb = WRITE_CB(write)
        self._check(self.lib.vss_save(self.h, cb, None))
        return state["off"]

    def load(self, blob):
        state = {"off": 0}

        def read(ctx, data, size):
            if state["off"] + size > len(blob):
                return 0
            C.memmove(data, blob[state["off"]:state["off"] + size], size)
            state["off"] += size
            return 1
        cb = READ_CB(read)
        self._check(self.lib.vss_load(self.h, cb, None))
        self.dim = self.lib.vss_dimensions(self.h)


def distance_batch(fn, a, b, device=0):
    """array_distance / array_cosine_distance / array_negative_inner_product over host arrays."""
    lib = load_library()
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    rows, dim = a.shape
    b_const = int(b.ndim == 1)
    out = np.zeros(rows, dtype=np.float32)
    rc = lib.vss_distance_batch(FUNCTIONS[fn], _p(a), _p(b), b_const, rows, dim, _p(out), device)
    if rc != 0:
        raise VssError("vss_distance_batch failed (no HIP device? the engine has no CPU fallback)")
    return out
