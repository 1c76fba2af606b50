"""Stage-by-stage bring-up diagnostics for the GPU box (not a pytest module).

    python tools/gpu_diag.py [stage ...]        # every stage runs in its own subprocess with a timeout

Prints one verdict line per stage plus details of the first mismatch, and rough timings at moderate sizes.
"""
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import gpu_common as gc  # noqa: E402
import datagen  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def stage_array():
    for dim in (3, 128, 768):
        A = datagen.normals(1, (500, dim)).astype(np.float32)
        B = datagen.normals(2, (500, dim)).astype(np.float32)
        got = gc.pkg().distance_batch("array_distance", A, B)
        ref = np.sqrt(((A.astype(np.float64) - B) ** 2).sum(1))
        print("  array_distance dim", dim, "max rel err", float(np.max(np.abs(got - ref) / ref)))


def _search_compare(n, dim, metric, ef=64, k=10):
    X, Q = gc.make_data(n, dim, metric, 100 + dim)
    cpu = gc.oracle_index(dim, metric)
    cpu.reserve(n)
    cpu.add_many(np.arange(n), X)
    gpu = gc.gpu_index(dim, metric)
    gpu.load(cpu.save())
    print("  load/save roundtrip identical:", gpu.save() == cpu.save())
    gk, gd, gcnt = gpu.search_batch(Q, k, ef)
    ck, cd, ccnt, cst = cpu.search_many(Q, k, ef=ef)
    ok = np.array_equal(gk, ck) and np.array_equal(bits(gd), bits(cd))
    print("  search n=%d dim=%d %s: keys %s dist-bits %s counts %s stats %s" % (
        n, dim, metric, np.array_equal(gk, ck), np.array_equal(bits(gd), bits(cd)), np.array_equal(gcnt, ccnt),
        np.array_equal(gpu.last_query_stats(len(Q)), cst.astype(np.uint32))))
    if not ok:
        bad = np.nonzero((gk != ck).any(1) | (bits(gd) != bits(cd)).any(1))[0]
        i = int(bad[0])
        print("   first bad query", i, "of", len(bad))
        print("   gpu", gk[i].tolist(), gd[i].tolist())
        print("   cpu", ck[i].tolist(), cd[i].tolist())
        print("   stats gpu", gpu.last_query_stats(len(Q))[i].tolist(), "cpu", cst[i].tolist())


def stage_search_small():
    _search_compare(500, 8, "l2sq")
    _search_compare(2000, 16, "cosine")
    _search_compare(2000, 20, "ip")


def stage_search_dims():
    _search_compare(3000, 128, "l2sq")
    _search_compare(1500, 768, "cosine")
    _search_compare(800, 1536, "l2sq")
    _search_compare(1200, 100, "l2sq")
    _search_compare(729, 3, "l2sq", ef=64, k=3)


def _build_compare(n, dim, metric, M, M0, efc, mb, gd_):
    X, Q = gc.make_data(n, dim, metric, 300 + n + dim)
    cpu = gc.oracle_index(dim, metric, M, M0, efc)
    cpu.reserve(n)
    t0 = time.time()
    cpu.build_batch(np.arange(n), X, mb, gd_)
    t1 = time.time()
    gpu = gc.gpu_index(dim, metric, M, M0, efc)
    gpu.reserve(n)
    gpu.set_build_params(mb, gd_)
    gpu.stage(np.arange(n), X)
    t2 = time.time()
    gpu.build_finalize()
    t3 = time.time()
    diff = gc.first_graph_difference(gpu.save(), cpu.save())
    print("  build n=%d dim=%d %s M=%d/%d efc=%d batch=%d/%d: %s   (cpu %.2fs gpu %.2fs)" % (
        n, dim, metric, M, M0, efc, mb, gd_, "IDENTICAL" if diff is None else "DIFF " + diff, t1 - t0, t3 - t2))


def stage_build_singleton():
    _build_compare(300, 8, "l2sq", 4, 8, 24, 1, 1)
    _build_compare(1000, 16, "l2sq", 16, 32, 128, 1, 1)


def stage_build_batched():
    _build_compare(3000, 16, "l2sq", 16, 32, 128, 256, 8)
    _build_compare(3000, 24, "cosine", 8, 16, 64, 512, 4)
    _build_compare(1200, 768, "l2sq", 16, 32, 128, 256, 8)


def stage_exact():
    n, dim = 5000, 64
    X, Q = gc.make_data(n, dim, "l2sq", 40 + dim, nq=33)
    cpu, gpu = gc.oracle_index(dim, "l2sq"), gc.gpu_index(dim, "l2sq")
    cpu.reserve(n), gpu.reserve(n)
    cpu.build_batch(np.arange(n), X, 256, 8)
    gpu.set_build_params(256, 8)
    gpu.add(np.arange(n), X)
    gk, gd, gcnt = gpu.search_batch(Q, 10, exact=True)
    ck, cd, ccnt, _ = cpu.search_many(Q, 10, exact=True)
    print("  exact: keys", np.array_equal(gk, ck), "dist-bits", np.array_equal(bits(gd), bits(cd)))
    if not np.array_equal(gk, ck):
        i = int(np.nonzero((gk != ck).any(1))[0][0])
        print("   q", i, gk[i].tolist(), ck[i].tolist(), gd[i].tolist(), cd[i].tolist())


def stage_perf():
    import torch
    for (n, dim, metric) in ((100_000, 128, "l2sq"), (100_000, 768, "cosine")):
        X, Q = gc.make_data(n, dim, metric, 555, nq=1024)
        gpu = gc.gpu_index(dim, metric)
        gpu.reserve(n)
        gpu.stage(np.arange(n), X)
        t0 = time.time()
        gpu.build_finalize()
        t1 = time.time()
        print("  build %dx%d %s: %.2fs = %.0f rows/s" % (n, dim, metric, t1 - t0, n / (t1 - t0)))
        ek, _, _ = gpu.search_batch(Q, 10, exact=True)
        t0 = time.time()
        ek, _, _ = gpu.search_batch(Q, 10, exact=True)
        t1 = time.time()
        print("  exact 1024 queries: %.3fs" % (t1 - t0))
        for ef in (64, 128, 256):
            gpu.search_batch(Q, 10, ef)
            t0 = time.time()
            for _ in range(5):
                k, d, c = gpu.search_batch(Q, 10, ef)
            t1 = time.time()
            st = gpu.last_search_stats()
            print("  search ef=%d: %.0f qps (host-pointer API), recall %.3f, dists/query %.0f, expansions/query %.1f, retried %d" % (
                ef, 5 * 1024 / (t1 - t0), gc.recall_at_k(k, ek), st[0] / 1024, st[1] / 1024, st[3]))
        del gpu
        torch.cuda.empty_cache()


def stage_single():
    """BASELINE configs[1]: 1M rows FLOAT[128] l2sq top-10, single-query HNSW_INDEX_SCAN entry point (vss_search)."""
    import torch
    n, dim = 1_000_000, 128
    X = datagen.mixture(n, dim, 2024, intrinsic_dim=32, basis_seed=2024, centre_scale=0.1)
    Q = datagen.mixture(2000, dim, 2025, n_clusters=1000, intrinsic_dim=32, basis_seed=2024, centre_scale=0.1)
    gpu = gc.gpu_index(dim, "l2sq")
    gpu.reserve(n)
    t0 = time.time()
    for c in range(0, n, 100_000):
        gpu.stage(np.arange(c, c + 100_000), X[c:c + 100_000])
    gpu.build_finalize()
    print("  build 1M x 128 l2sq (reference defaults M=16): %.2fs incl. host->device staging" % (time.time() - t0))
    ek, _, _ = gpu.search_batch(Q, 10, exact=True)
    for ef in (64, 128):
        for q in Q[:50]:
            gpu.search(q, 10, ef)
        t0 = time.time()
        got = [gpu.search(q, 10, ef) for q in Q]
        dt = time.time() - t0
        rec = np.mean([len(set(got[i].tolist()) & set(ek[i].tolist())) / 10 for i in range(len(Q))])
        kms = []
        for q in Q[:200]:
            gpu.search(q, 10, ef)
            kms.append(gpu.timing()["search_kernel_ms"])
        print("  single-query vss_search ef=%d: %.0f queries/s (%.1f us per call, through ctypes; kernel alone %.1f us mean), "
              "recall@10 %.3f" % (ef, len(Q) / dt, dt / len(Q) * 1e6, float(np.mean(kms)) * 1e3, rec))
        bk, _, _ = gpu.search_batch(Q, 10, ef)
        t0 = time.time()
        bk, _, _ = gpu.search_batch(Q, 10, ef)
        print("  same 2000 queries as ONE batch (host pointers): %.0f queries/s" % (len(Q) / (time.time() - t0)))


def stage_array_bw():
    """array_* scalar functions over a device-resident column: streaming HBM bandwidth."""
    import torch
    lib = gc.pkg().load_library()
    rows, dim = 4_000_000, 768
    a = torch.randn(rows, dim, device="cuda")
    b = torch.randn(rows, dim, device="cuda")
    c = torch.randn(dim, device="cuda")
    out = torch.empty(rows, device="cuda")
    torch.cuda.synchronize()
    for name, fn in (("array_distance", 0), ("array_cosine_distance", 1), ("array_negative_inner_product", 2)):
        for label, bb, const in (("column x constant", c, 1), ("column x column", b, 0)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            lib.vss_distance_batch_device(fn, a.data_ptr(), bb.data_ptr(), const, rows, dim, out.data_ptr(), None)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                lib.vss_distance_batch_device(fn, a.data_ptr(), bb.data_ptr(), const, rows, dim, out.data_ptr(), None)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            gb = (rows * dim * 4 * (1 if const else 2) + rows * 4) / 1e9
            print("  %-30s %-18s %.3f ms  %.0f GB/s" % (name, label, ms, gb / (ms / 1e3)))
    ref = torch.sqrt(((a[:1000] - c) ** 2).sum(1))
    lib.vss_distance_batch_device(0, a.data_ptr(), c.data_ptr(), 1, 1000, dim, out.data_ptr(), None)
    torch.cuda.synchronize()
    print("  max rel err vs torch:", float(((out[:1000] - ref).abs() / ref).max()))


STAGES = {
    "array": stage_array,
    "search_small": stage_search_small,
    "search_dims": stage_search_dims,
    "build_singleton": stage_build_singleton,
    "build_batched": stage_build_batched,
    "exact": stage_exact,
    "perf": stage_perf,
    "single": stage_single,
    "array_bw": stage_array_bw,
}

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--run":
        STAGES[sys.argv[2]]()
        sys.exit(0)
    names = sys.argv[1:] or list(STAGES)
    for name in names:
        t0 = time.time()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--run", name], capture_output=True, text=True,
                               timeout=int(os.environ.get("DIAG_TIMEOUT", "240")))
            verdict = "exit %d" % p.returncode
            out = p.stdout + ("\n" + p.stderr[-3000:] if p.returncode else "")
        except subprocess.TimeoutExpired as e:
            verdict = "TIMEOUT"
            out = (e.stdout or b"").decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        print("=== stage %s: %s (%.1fs)" % (name, verdict, time.time() - t0))
        print(out)
        sys.stdout.flush()
