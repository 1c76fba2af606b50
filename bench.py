#!/usr/bin/env python3
"""bench.py — the headline benchmark of BASELINE.json on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 10 000 000 rows x FLOAT[768], metric
cosine, top-10, batches of 1024 queries, one MI355X.  A "step" is ONE pass of the hot path over one batch: one
HNSW_INDEX_JOIN-style probe of 1024 queries (vss_search_batch_device) against the resident index, inputs already in
HBM.  Reported: queries/s at the recall@10 measured on HELD-OUT batches against the exact MFMA brute-force path (ef_search =
the smallest of EF_SWEEP whose recall on two selection batches clears 0.95 by two standard errors), plus index build rows/s,
the HBM roofline of the search kernel and a CPU baseline (the reference's own usearch build, one thread, on the very graph
the engine built) whose answers must agree with the engine's cell by cell: a differing cell has to be a near-tie in the
reference's arithmetic, or its query must replay bit-identically in the oracle's wave-order restatement — else the run fails.

The default single-GPU run then releases its index and attaches, as objects of the same JSON line, the other BASELINE
configurations in compact form — c2 (configs[1]), c4 (configs[3] on one GPU), c5 (one shard of configs[4]), a13 (the three
array_* functions) and the headline workload on the reference's DEFAULT index options — each measured by `bench.py --config
...` in a process of its own (`--extras none` = just the headline).

N > 1 (launched by torch.distributed.run, one rank per GPU): BASELINE.json configs[3] — the same 10M x 768 rows (l2sq)
row-range sharded across the ranks, every rank answers every query on its shard, ONE RCCL all-gather per launch of the
search engine carrying the per-shard top-k of all its batches (packed ids + distances), k-way merge kernel.  Total work
is fixed -> "scaling": "strong".

`--config c4` runs configs[3] at its full workload on ONE GPU: the 10M x 768 l2sq rows as 8 row-range shards co-resident
on the device (30.7 GB), every shard answers every batch, the per-shard results are merged by the same packed merge
kernel — no collective, no xGMI; what the 8-GPU run does minus the exchange.

Data: synthetic, seeded Gaussian mixture with low intrinsic dimension: sqrt(N) centres ~ 0.1 * N(0, I)
(overlapping clusters: with well separated ones a descent that lands in the wrong cluster cannot leave it), row = centre + 0.3 * (z @ B), z ~ N(0, I_32), B a fixed
32 x dim basis with unit-norm-ish rows; L2-normalised for cosine.  Queries come from the same mixture with a disjoint seed.

`--config c2` runs BASELINE.json configs[1] instead (1M rows x FLOAT[128], l2sq, reference default options, top-10, ONE
query per call through the HNSW_INDEX_SCAN entry point vss_search): a step is one query, `value` is single-query
queries/s, the roofline object is latency-bound (microseconds per graph expansion) and the CPU baseline is the reference
library answering the same queries one by one on the same graph.

Development overrides (NOT the benchmark): --rows / --dim / --metric shrink the workload for quick runs; the JSON
line then says so in config.workload.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from __graft_entry__ import load_package  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# index options of the headline run (WITH (M = 32, ef_construction = 384); M0 = 2 M as the reference derives it,
# hnsw_index.cpp:208-217) and the ef_search values its operating point is chosen from — also what
# tests/test_gpu_parity.py::test_properties_at_full_benchmark_size runs
HEADLINE_OPTIONS = {"M": 32, "ef_construction": 384}
EF_SWEEP = [16, 24, 32, 40, 48, 52, 56, 60, 64, 68, 72, 76, 80, 84, 88, 92, 96, 100, 104, 112, 120, 128, 144, 160, 192, 224, 256,
            320, 384, 448, 512]
EF_SWEEP_WIDE = [640, 768, 1024, 1536]  # --wide-ef-sweep (the reference-default index: limits beyond the register lists)
DATA_SEED, QUERY_SEED = 0xD0C5EED, 0x5EEDBEEF
INTRINSIC_DIM = int(os.environ.get("VSS_BENCH_INTRINSIC_DIM", 32))  # experiments only
SPREAD, CENTRE_SCALE = 0.3, 0.1
CHUNK = 500_000


class Mixture:
    """Seeded generator: any row chunk can be regenerated from (seed, chunk index)."""

    def __init__(self, n_total, dim, normalize, device):
        self.dim, self.normalize, self.device = dim, normalize, device
        g = torch.Generator(device=device).manual_seed(DATA_SEED)
        self.k = max(2, int(math.sqrt(n_total)))
        self.centres = CENTRE_SCALE * torch.randn(self.k, dim, generator=g, device=device)
        self.basis = torch.randn(INTRINSIC_DIM, dim, generator=g, device=device) / math.sqrt(dim)

    def rows(self, seed, chunk_index, n):
        g = torch.Generator(device=self.device).manual_seed(seed + 7919 * chunk_index)
        assign = torch.randint(0, self.k, (n,), generator=g, device=self.device)
        z = torch.randn(n, INTRINSIC_DIM, generator=g, device=self.device)
        x = self.centres[assign] + SPREAD * (z @ self.basis)
        if self.normalize:
            x = x / x.norm(dim=1, keepdim=True)
        return x.contiguous()


def recall_at_k(got, truth):
    g, t = got.cpu().numpy(), truth.cpu().numpy()
    k = t.shape[1]
    return float(np.mean([len(set(g[i].tolist()) & set(t[i].tolist())) / k for i in range(len(t))]))


def agreement(cpu_keys, cpu_d, gpu_keys, gpu_d, metric, ef, queries=None, fetch_rows=None, ref_distance=None, replay=None):
    """Reference answers against the engine's for the same queries on the same graph at the same ef_search (what
    HNSWIndex::InitializeScan returns: index.ef_search(...) + dump_to, reference hnsw_index.cpp:333-339): fraction of
    (query, rank) cells naming the same row, and the largest relative difference of the distances of those cells (the
    two sides sum in different orders: north_star's bar is 1e-5; for cosine / ip the distance is 1 - s, so the error is
    taken relative to max(|d|, |1 - d|) as in tests/test_gpu_parity2.py).

    Round 4 — identical modulo near-ties, cell by cell: for every cell whose row ids differ, both rows are fetched and their
    distances to the query are recomputed with ONE arithmetic, the reference's own metric in the reference's summation order
    (`ref_distance` = orc_distance of the library the baseline runs on).  The cell is explained iff the two distances differ
    by at most 1e-5 relative — the two rows are a near-tie that the engine's wave-order sums and the reference's sequential
    sums may legitimately rank either way.  What is left — a row one side did not return at all (every later rank of that
    query then holds a different pair), a gap beyond 1e-5 — can still be the two arithmetics parting ways at a near-tie DEEP
    in the traversal (a candidate admitted by one side's `d < radius` and not by the other's changes what gets expanded).
    That is not taken on trust: `replay(query indices)` answers those very queries once more with the oracle in kernel mode —
    the CPU restatement of the traversal in the ENGINE's summation order, on the same graph — and a query counts as explained
    only if that replay returns the engine's answer bit for bit (ids and distance bits): the engine then took the reference's
    decisions, in wave-order arithmetic.  Whatever survives both checks is in `unexplained_mismatches`, and a run with
    unexplained mismatches is a FAILED run."""
    n = min(len(cpu_keys), len(gpu_keys))
    ck, gk = np.asarray(cpu_keys[:n]), np.asarray(gpu_keys[:n])
    cd, gd = np.asarray(cpu_d[:n], dtype=np.float64), np.asarray(gpu_d[:n], dtype=np.float64)
    same = ck == gk

    def scale(d):
        return np.maximum(np.abs(d), 1e-30) if metric == "l2sq" else np.maximum(np.abs(d), np.abs(1.0 - d))
    rel = np.abs(gd - cd) / scale(cd)
    out = {"queries": int(n), "ef_search": int(ef), "id_match_frac": float(same.mean()) if n else None,
           "query_match_frac": float(same.all(axis=1).mean()) if n else None,
           "rank_distance_max_rel_err": float(rel[same].max()) if same.any() else None,
           "rank_distance_max_rel_err_all_cells": float(rel.max()) if n else None,
           "distance_error_relative_to": "d" if metric == "l2sq" else "max(|d|, |1-d|)",
           "mismatching_cells": int((~same).sum()), "unexplained_mismatches": None,
           "bars": {"id_match_frac_min": 0.99, "rank_distance_max_rel_err_max": 1e-5, "unexplained_mismatches_max": 0}}
    if queries is not None and fetch_rows is not None and ref_distance is not None:
        cells = np.argwhere(~same)
        keys = sorted({int(x) for x in np.concatenate([ck[~same], gk[~same]]) if x >= 0}) if len(cells) else []
        rows = fetch_rows(keys) if keys else {}
        worst, open_cells = 0.0, []  # open_cells: not a near-tie of the two rows in the cell
        for i, r in cells:
            a, b = int(ck[i, r]), int(gk[i, r])
            da = db = None
            if a >= 0 and b >= 0 and a in rows and b in rows:
                da, db = float(ref_distance(queries[i], rows[a])), float(ref_distance(queries[i], rows[b]))
                gap = abs(da - db) / float(scale(np.float64(da)))
                if gap <= 1e-5:
                    worst = max(worst, gap)
                    continue
            open_cells.append((int(i), int(r), a, b, da, db))
        replayed = {}
        if open_cells and replay is not None:
            qs = sorted({c[0] for c in open_cells})
            rk, rd = replay(qs)
            for j, qi in enumerate(qs):  # the wave-order restatement must return the ENGINE's answer, bit for bit
                replayed[qi] = bool(np.array_equal(np.asarray(rk[j]), gk[qi]) and
                                    np.array_equal(np.asarray(rd[j], dtype=np.float32).view(np.uint32),
                                                   np.asarray(gpu_d[qi], dtype=np.float32).view(np.uint32)))
        left = [c for c in open_cells if not replayed.get(c[0], False)]
        out.update({"unexplained_mismatches": len(left), "mismatch_max_rel_distance_gap": worst,
                    "cells_beyond_a_near_tie": len(open_cells),
                    "queries_replayed_in_wave_order": len(replayed), "queries_replay_identical_to_engine": sum(replayed.values()),
                    "mismatch_check": "every cell whose ids differ: both rows fetched, distances to the query recomputed in the "
                                      "reference's arithmetic (orc_distance), explained iff they differ by <= 1e-5 relative; the "
                                      "queries of the remaining cells replayed by the oracle in kernel mode (the traversal "
                                      "restated in the engine's summation order, same graph): explained iff that replay equals "
                                      "the engine's answer bit for bit"})
        if left:
            out["unexplained_examples"] = [{"query": c[0], "rank": c[1], "reference_row": c[2], "engine_row": c[3],
                                            "reference_order_distances": [c[4], c[5]]} for c in left[:4]]
    return out


def agreement_ok(a):
    return (a is not None and a["queries"] > 0 and a["id_match_frac"] >= 0.99 and
            a["rank_distance_max_rel_err"] is not None and a["rank_distance_max_rel_err"] <= 1e-5 and
            not a.get("unexplained_mismatches"))


def make_row_fetch(gen, total_rows, full_chunks=True):
    """key -> row for data staged chunk by chunk from the generator with row id = global row number: the chunk that holds a
    key is regenerated on the device (the generator is counter-based: same seed, same chunk index, same length -> same
    values) and the wanted rows are copied to the host.  full_chunks: every chunk was generated CHUNK rows long and cut
    (main, c4); otherwise the last chunk was generated at its own length (c2)."""
    def fetch(keys):
        out, by_chunk = {}, {}
        for key in keys:
            by_chunk.setdefault(int(key) // CHUNK, []).append(int(key))
        for ci, ks in by_chunk.items():
            n = CHUNK if full_chunks else min(CHUNK, total_rows - ci * CHUNK)
            x = gen.rows(DATA_SEED, ci, n)
            sel = torch.tensor([key - ci * CHUNK for key in ks], device=x.device)
            got = x[sel].cpu().numpy()
            for key, row in zip(ks, got):
                out[key] = np.ascontiguousarray(row, dtype=np.float32)
            del x
        return out
    return fetch


def make_ref_distance(lib, metric, dim):
    """The baseline library's own metric in its own summation order (orc_distance: index_plugins.hpp:977-1053 restated by the
    oracle, or the reference's metric_punned_t itself through oracle/ref_shim.cpp)."""
    from oracle_lib import METRICS as ORC_METRICS

    def ref_distance(a, b):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        return lib.orc_distance(ORC_METRICS[metric], a.ctypes.data, b.ctypes.data, dim)
    return ref_distance


def make_wave_order_replay(streams, dim, metric, M, M0, efc, k, ef, queries):
    """replay(query indices) for agreement(): the oracle (oracle/hnsw_oracle.cpp, TEST INFRASTRUCTURE — used here as the checker
    of the cpu_baseline leg, never as the thing measured) in kernel mode — order = 1 (the engine's wave summation order), wave
    = 1 (the engine's candidate list) — loads the same graph(s) through the stream format and answers the given queries;
    several graphs (row-range shards) are merged by (distance, row id) like the engine's merge.  `streams()` yields the
    serialized graphs one at a time (called only if a replay is needed: the reference indexes have been released by then)."""
    def replay(idxs):
        from oracle_lib import CpuIndex, load_oracle
        lib = load_oracle()
        parts = []
        for buf, length in streams():
            orc = CpuIndex(lib, dim, metric, M, M0, efc, 64, order=1, wave=1)
            orc.load_buffer(buf, length)
            del buf
            parts.append([orc.search(queries[i], k, ef=ef)[:2] for i in idxs])
            del orc
        keys, ds = [], []
        for j in range(len(idxs)):
            kk = np.concatenate([p[j][0] for p in parts])
            dd = np.concatenate([p[j][1] for p in parts])
            order = np.lexsort((kk, dd))[:k] if len(parts) > 1 else np.arange(min(k, len(kk)))
            kj, dj = np.full(k, -1, dtype=np.int64), np.full(k, np.inf, dtype=np.float32)
            kj[:len(order)], dj[:len(order)] = kk[order], dd[order]
            keys.append(kj)
            ds.append(dj)
        return keys, ds
    return replay


def stream_of(ix):
    """(buffer, length) of one index's serialized graph (the reference stream format)."""
    buf = np.empty(ix.serialized_length(), dtype=np.uint8)
    return buf, ix.save_into(buf)


def visited_set_form(limit, rows_per_index):
    """Where a walker's exact visited set lives at this search limit (= max(k, ef_search)) — the engine's sizing rule
    (csrc/host_logic.h: search_cells_per_limit, compact_visited_cells_log2), restated for the JSON line.  Indexes
    so small that the plain table fits LDS anyway keep it there whatever the limit."""
    if limit <= 128:
        return "32-bit cells in LDS (64 per entry of the limit)"
    if limit <= 256:
        return "32-bit cells in LDS (32 per entry of the limit; queries that outgrow it are re-run with more)"
    if limit <= 512 and rows_per_index <= 1 << 25 and os.environ.get("VSS_VISITED_COMPACT", "1") != "0":
        return "16-bit cells (tag + displacement, exact) in LDS; a set that outgrows them moves to 32-bit cells in HBM"
    return "32-bit cells in HBM"


def mean_and_se(values):
    v = np.asarray(values, dtype=np.float64)
    return float(v.mean()), float(v.std(ddof=1) / math.sqrt(len(v))) if len(v) > 1 else 0.0


def recall_per_query(got, truth):
    g, t = got.cpu().numpy(), truth.cpu().numpy()
    k = t.shape[1]
    return [len(set(g[i].tolist()) & set(t[i].tolist())) / k for i in range(len(t))]


def select_ef(recalls_at, sweep, target):
    """The operating point of the benchmark: the smallest ef_search of the sweep whose recall on the SELECTION queries clears
    the target by two standard errors of that estimate (so that the choice is not an artefact of the sample it was made on;
    round 3 picked the first ef at >= 0.95 in-sample and reported that same number).  `recalls_at(ef)` returns the per-query
    recall of the selection queries.  Returns (ef, mean, standard error, log); the recall the bench REPORTS is then measured
    on held-out batches the selection never saw."""
    log, ef, mean, se = [], sweep[-1], 0.0, 0.0
    for e in sweep:
        mean, se = mean_and_se(recalls_at(e))
        log.append({"ef": e, "recall": round(mean, 4), "se": round(se, 5)})
        ef = e
        if mean - 2.0 * se >= target:
            break
    return ef, mean, se, log


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pkg, args, gen, dim, metric, k, ef, device, M, M0, efc, shards=None, gpu_answer=None):
    """The reference path on this box's host cores, 1 thread (HNSW_INDEX_SCAN / HNSW_INDEX_JOIN are single-threaded
    operators: reference hnsw_index_scan.cpp:172, hnsw_optimize_join.cpp:65-67).  Bounded sample: the very graph(s) the
    engine built and was measured on, handed to the CPU library through the reference's stream format (or, when host RAM
    does not allow, a prefix index built with identical parameters), then the same queries searched one by one for
    ~args.cpu_seconds.  With several shards a query is answered by every shard's graph in turn and merged by (distance,
    row id) — the CPU equivalent of the sharded probe.  The answers of the first pass are KEPT and compared with the
    engine's answers to the same queries (`gpu_answer`): cpu_baseline.agreement."""
    from oracle_lib import CpuIndex, load_oracle, load_ref
    lib, kind = load_ref(), "reference"
    if lib is None:
        lib, kind = load_oracle(), "port"
    shards = shards or []
    sample_rows, sample_what = None, None
    need = sum(ix.serialized_length() for ix in shards)
    biggest = max([ix.serialized_length() for ix in shards] + [0])
    avail = 0
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable"):
                    avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    cpus, prefix_gpu = [], None
    fetch_rows = make_row_fetch(gen, args.rows, full_chunks=True)
    if shards and not args.cpu_prefix_only and avail > 2 * need + biggest + (16 << 30):
        for ix in shards:  # same graphs, same data
            cpu = CpuIndex(lib, dim, metric, M, M0, efc, 64)
            buf = np.empty(ix.serialized_length(), dtype=np.uint8)
            cpu.load_buffer(buf, ix.save_into(buf))
            del buf
            cpus.append(cpu)
        sample_rows = args.rows
        sample_what = ("the full %d-row index built by the engine" % args.rows if len(shards) == 1 else
                       "the %d row-range shard graphs (%d rows in all) built by the engine, every query answered by each and "
                       "merged" % (len(shards), args.rows))
    else:
        sample_rows = min(args.rows, args.cpu_sample_rows)
        x = gen.rows(DATA_SEED, 0, sample_rows)
        ids = torch.arange(sample_rows, dtype=torch.int64, device=device)
        prefix_gpu = pkg.GpuIndex(dim, metric, M, M0, efc, 64, device=device.index or 0)
        prefix_gpu.reserve(sample_rows)
        prefix_gpu.stage_device(ids.data_ptr(), x.data_ptr(), sample_rows)
        prefix_gpu.build_finalize()
        cpu = CpuIndex(lib, dim, metric, M, M0, efc, 64)
        cpu.load(prefix_gpu.save())
        cpus.append(cpu)
        prefix_rows = x  # (generated in one call of its own length: rows are fetched from this very tensor)
        fetch_rows = lambda keys: {int(key): prefix_rows[int(key)].cpu().numpy() for key in keys}  # noqa: E731
        gpu_answer = lambda qs: prefix_gpu.search_batch(qs, k, ef)[:2]  # noqa: E731 (the prefix graph is the one compared)
        sample_what = "a %d-row prefix index of the same data (graph built with identical parameters)" % sample_rows

    def cpu_search(qv):
        if len(cpus) == 1:
            kk, dd, _ = cpus[0].search(qv, k, ef=ef)
            return kk, dd
        parts = [c.search(qv, k, ef=ef) for c in cpus]
        kk = np.concatenate([p[0] for p in parts])
        dd = np.concatenate([p[1] for p in parts])
        order = np.lexsort((kk, dd))[:k]  # ascending (distance, row id): dump_to's order over the union
        return kk[order], dd[order]

    q = gen.rows(QUERY_SEED, 0, 4096).cpu().numpy()
    for i in range(64 if len(cpus) == 1 else 8):  # warm-up (page in the index, settle the clock)
        cpu_search(q[i])
    # three timed windows over the same query stream; the best one is reported (the host is shared)
    n_keep = 2048
    kept_k = np.full((n_keep, k), -1, dtype=np.int64)
    kept_d = np.full((n_keep, k), np.inf, dtype=np.float32)
    done, search_s, rates, pos = 0, 0.0, [], 0
    for _ in range(3):
        t0 = time.perf_counter()
        n_win = 0
        while time.perf_counter() - t0 < args.cpu_seconds / 3:
            kk, dd = cpu_search(q[pos % len(q)])
            if pos < n_keep:
                kept_k[pos, :len(kk)], kept_d[pos, :len(dd)] = kk, dd
            pos += 1
            n_win += 1
        dt = time.perf_counter() - t0
        rates.append(n_win / dt)
        done += n_win
        search_s += dt
    # build rate: sequential add() of a small prefix into a fresh CPU index (small graph: favours the CPU)
    xb = gen.rows(DATA_SEED, 0, 20000).cpu().numpy()
    cb = CpuIndex(lib, dim, metric, M, M0, efc, 64)
    cb.reserve(len(xb), 1)
    t0 = time.perf_counter()
    nb = 0
    while nb < len(xb) and time.perf_counter() - t0 < args.cpu_seconds:
        cb.add(nb, xb[nb])
        nb += 1
    build_s = time.perf_counter() - t0
    # All host cores, the reference's own threading model (reference build only): concurrent searches with one context per
    # thread = the upper bound for concurrent sessions; the bulk build with one add() stream per scheduler thread is what
    # CREATE INDEX actually runs (hnsw_index_physical_create.cpp:239-245).
    all_cores = None
    if kind == "reference" and len(cpus) == 1:
        try:
            threads = len(os.sched_getaffinity(0))
        except AttributeError:
            threads = os.cpu_count() or 1
        budget = args.cpu_seconds / 2  # each leg stops handing out work after this many seconds
        s_mt, n_mt, _ = cpus[0].search_mt(q, k, ef, threads, 1 << 40, budget)
        nb_mt = min(args.rows, args.cpu_mt_build_rows)
        xm = gen.rows(DATA_SEED, 0, nb_mt).cpu().numpy()
        cm = CpuIndex(lib, dim, metric, M, M0, efc, 64)
        s_build, n_build = cm.add_mt(np.arange(nb_mt), xm, threads, budget)
        quota = None  # a container CPU quota (cgroup v2) caps what "all cores" can deliver
        try:
            with open("/sys/fs/cgroup/cpu.max") as f:
                quota = f.read().strip()
        except OSError:
            pass
        all_cores = {"threads": threads, "cgroup_cpu_max": quota, "search_queries_per_s": n_mt / s_mt, "search_queries": n_mt,
                     "build_rows_per_s": n_build / s_build, "build_rows": n_build,
                     "note": "search: same full index and queries, one usearch context per thread; build: rows added to an "
                             "empty index in %.0f s, one add() stream per thread over 2048-row chunks (a small graph favours "
                             "the CPU)" % budget}
        del cm, xm
    # agreement of the kept answers with the engine's — last, so that the reference indexes can be released before a
    # wave-order replay (needed only if some cell is not a plain near-tie) loads the same graphs into the oracle
    agree = None
    if gpu_answer is not None:
        n_cmp = min(pos, n_keep)
        gk, gd = gpu_answer(q[:n_keep])
        graphs = [prefix_gpu] if prefix_gpu is not None else shards
        del cpus[:]

        def streams():
            for ix in graphs:
                buf = np.empty(ix.serialized_length(), dtype=np.uint8)
                yield buf, ix.save_into(buf)
        agree = agreement(kept_k[:n_cmp], kept_d[:n_cmp], gk[:n_cmp], gd[:n_cmp], metric, ef, queries=q[:n_cmp],
                          fetch_rows=fetch_rows, ref_distance=make_ref_distance(lib, metric, dim),
                          replay=make_wave_order_replay(streams, dim, metric, M, M0, efc, k, ef, q))
    if prefix_gpu is not None:
        prefix_gpu.close()
        prefix_rows = None
    return {
        "value": max(rates), "unit": "queries/s", "cores": 1, "kind": kind, "window_rates": [round(r, 1) for r in rates],
        "sample": "best of 3 windows, %d single-thread ef_search(k=%d, ef=%d) calls in total, on %s, loaded via the "
                  "reference stream format; build: %d sequential add() calls into an empty index" % (
                      done, k, ef, sample_what, nb),
        "index_rows": sample_rows, "agreement": agree,
        "build_rows_per_s": nb / build_s, "host_cores_available": os.cpu_count(), "cpu_model": cpu_model_name(),
        "all_cores": all_cores,
    }


# ------------------------------------------------------------------------------------------------ what gets printed
# The driver keeps a bounded tail of stdout (8 000 characters) and parses the LAST JSON line of it.  Round 4 lost its
# headline to a 24 kB line.  The contract since round 5 (tests/test_bench_helpers.py holds it):
#   * the LAST stdout line is the compact headline object, at most LINE_LIMIT characters;
#   * everything else — ef sweep, launch regimes, latency shapes, the host API leg, each extra configuration — is printed
#     on EARLIER lines of its own, each a small JSON object starting with {"detail": ...} or {"extra": ...}, short enough
#     that all of them together with the headline fit the driver's tail;
#   * the complete, unabridged result goes to a sidecar file (--sidecar, default gpurun_out/bench_full_<config>.json).
LINE_LIMIT = 3500      # the last line
SIDE_LINE_LIMIT = 540  # every {"detail": ...} / {"extra": ...} line
AGREEMENT_SCALARS = ("queries", "id_match_frac", "query_match_frac", "rank_distance_max_rel_err", "mismatching_cells",
                     "unexplained_mismatches")
HEADLINE_KEYS = ("metric", "config_id", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "multi_gpu_mode", "vs_baseline", "dtype", "data", "recall_at_10", "recall_at_10_se", "recall_at_100",
                 "recall_at_100_se", "recall_measured_on", "ef_search", "repeat", "plateau", "build_rows_per_s", "build", "rccl_ranks",
                 "collective_backend", "collectives_per_launch", "collectives_timed", "rank_pci", "config", "roofline", "cpu_baseline",
                 "exit_code", "wall_s")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "traffic_source",
                 "algorithmic_bytes_per_launch", "avg_kernel_ms", "launches", "frac_over_wall", "distances_per_query",
                 "expansions_per_query", "visited_set", "latency_bound", "us_per_expansion", "slowest_leg")
CPU_BASELINE_KEYS = ("value", "unit", "cores", "kind", "cpu_model", "host_cores_available", "index_rows", "build_rows_per_s",
                     "sample", "agreement")
DROP_ORDER = (("cpu_baseline", "sample"), ("roofline", "traffic_source"), ("config", "workload_detail"), ("roofline", "visited_set"),
              ("build",), ("rank_pci",), ("recall_measured_on",), ("roofline", "slowest_leg"), ("cpu_baseline", "host_cores_available"),
              ("cpu_baseline", "index_rows"), ("roofline", "launches"), ("roofline", "frac_over_wall"), ("plateau",), ("repeat",))


def rounded(x, sig=6):
    """Floats to `sig` significant digits, recursively (a queries/s figure does not need 17 of them)."""
    if isinstance(x, float):
        return x if (x != x or x in (float("inf"), float("-inf")) or x == 0.0) else float("%.*g" % (sig, x))
    if isinstance(x, dict):
        return {k: rounded(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [rounded(v, sig) for v in x]
    return x


def clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + "…"


def compact_agreement(a):
    return None if a is None else {k: a.get(k) for k in AGREEMENT_SCALARS}


def compact_line(result, limit=LINE_LIMIT):
    """The headline object of the LAST stdout line: the contract's keys, `config`, `roofline` without its arrays, `cpu_baseline`
    with its agreement reduced to six scalars — never more than `limit` characters, whatever the field widths."""
    out = {k: result[k] for k in HEADLINE_KEYS if k in result}
    if isinstance(result.get("roofline"), dict):
        out["roofline"] = {k: clip(result["roofline"][k], 90) for k in ROOFLINE_KEYS if k in result["roofline"]}
    cb = result.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = {k: clip(cb[k], 150) for k in CPU_BASELINE_KEYS if k in cb}
        out["cpu_baseline"]["agreement"] = compact_agreement(cb.get("agreement"))
    if isinstance(result.get("config"), dict):
        out["config"] = {k: clip(v, 220) for k, v in result["config"].items() if not isinstance(v, (dict, list))}
    out["metric"] = clip(out.get("metric"), 160)
    out = rounded(out)
    for path in DROP_ORDER:  # worst-case field widths: shed the optional parts, least important first
        if len(json.dumps(out)) <= limit:
            break
        node = out
        for key in path[:-1]:
            node = node.get(key) if isinstance(node, dict) else None
        if isinstance(node, dict):
            node.pop(path[-1], None)
    if len(json.dumps(out)) > limit:  # cannot happen with the keys above; the contract's own keys are what must survive
        out = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                   "scaling", "vs_baseline", "dtype", "data") if k in out}
        out["metric"] = clip(out.get("metric"), 120)
        out["truncated"] = True
    return out


def extra_line(name, obj, limit=SIDE_LINE_LIMIT):
    """One compact line per extra configuration: what it measured, at what recall, how far from its roofline, whether the
    reference agreed — the process's own (already compact) last line is in the sidecar."""
    if "error" in obj:
        return {"extra": name, "error": clip(obj["error"], 200), "wall_s": obj.get("wall_s")}
    rl, cb = obj.get("roofline") or {}, obj.get("cpu_baseline") or {}
    cfg = obj.get("config") or {}
    out = {"extra": name, "value": obj.get("value"), "unit": obj.get("unit"),
           "recall": obj.get("recall_at_10", obj.get("recall_at_100")), "ef_search": obj.get("ef_search"),
           "build_rows_per_s": obj.get("build_rows_per_s"),
           "rows": cfg.get("rows"), "dim": cfg.get("dim"), "index_metric": cfg.get("index_metric"), "k": cfg.get("k"),
           "kernel": clip(rl.get("kernel"), 40), "frac": rl.get("frac"), "traffic_over_algorithmic": rl.get("traffic_over_algorithmic"),
           "avg_kernel_ms": rl.get("avg_kernel_ms"),
           "distances_per_query": rl.get("distances_per_query"), "expansions_per_query": rl.get("expansions_per_query"),
           "visited_set": clip(rl.get("visited_set"), 60), "us_per_expansion": rl.get("us_per_expansion"),
           "cpu_value": cb.get("value"), "cpu_kind": cb.get("kind"), "agreement": compact_agreement(cb.get("agreement")),
           "crud_recall_qps": [[c.get("recall_at_100", c.get("recall_at_10")), round(c.get("queries_per_s") or 0)]
                               for c in obj["crud"]] if obj.get("crud") else None,
           "chunk_2048_us": obj.get("chunk_2048_us"), "cpu_chunk_2048_us": obj.get("cpu_chunk_2048_us"),
           "crossover_rows": obj.get("crossover_rows"),
           "plateau": ({k_: v_ for k_, v_ in obj["plateau"].items() if k_ != "note"} if isinstance(obj.get("plateau"), dict) else None),
           "quality": obj.get("quality_compact"),
           "at_ef_512": ({k_: obj["at_ef_512"].get(k_) for k_ in ("queries_per_s", "frac", "recall_at_10_selection_batches")}
                         if isinstance(obj.get("at_ef_512"), dict) else None),
           "exit_code": obj.get("exit_code"), "wall_s": obj.get("wall_s")}
    out = rounded({k: v for k, v in out.items() if v is not None}, 5)
    for k in ("visited_set", "kernel", "avg_kernel_ms", "expansions_per_query", "distances_per_query", "build_rows_per_s", "unit",
              "index_metric", "k", "agreement", "plateau", "crud_recall_qps", "cpu_kind", "crossover_rows", "quality",
              "cpu_chunk_2048_us", "chunk_2048_us", "us_per_expansion", "cpu_value", "rows", "dim", "traffic_over_algorithmic"):
        if len(json.dumps(out)) <= limit:
            break
        out.pop(k, None)
    return out


def detail_lines(result, limit=SIDE_LINE_LIMIT):
    """The parts of a result that are lists or explanations, one small line each (printed BEFORE the headline)."""
    lines = []

    def add(name, body):
        line = rounded({"detail": name, **body}, 5)
        if len(json.dumps(line)) <= limit:
            lines.append(line)
        else:
            lines.append({"detail": name, "note": "too long for a side line: see the sidecar file"})
    if result.get("ef_sweep"):
        add("ef_sweep", {"ef:recall": {str(e["ef"]): e["recall"] for e in result["ef_sweep"]}})
    rec = result.get("recall")
    if isinstance(rec, dict):
        add("recall", {k: ({kk: vv for kk, vv in v.items() if kk != "rule"} if isinstance(v, dict) else v) for k, v in rec.items()})
    for r in (result.get("roofline") or {}).get("regimes") or []:
        add("regime", {"batches_per_launch": r.get("batches_per_launch"), "in_flight": r.get("launches_in_flight"),
                       "gated": r.get("gated"), "queries_per_s": r.get("queries_per_s"), "avg_kernel_ms": r.get("avg_kernel_ms"),
                       "frac_per_launch": r.get("frac_per_launch"), "frac_over_wall": r.get("frac_over_wall")})
    sm = result.get("small_launches")
    if isinstance(sm, dict):
        add("small_launches", {"ef_search": sm.get("ef_search"),
                               "single_query_us": sm["single_query"]["us_per_call"],
                               "single_query_reference_thread_us": sm["single_query"].get("reference_thread_us_per_call"),
                               "join_chunk_queries": sm["join_chunk"]["queries"], "join_chunk_us": sm["join_chunk"]["us_per_call"],
                               "join_chunk_reference_thread_us": sm["join_chunk"].get("reference_thread_us_per_call")})
    jc = result.get("join_chunk")
    if isinstance(jc, dict):
        add("join_chunk", {k: v for k, v in jc.items() if k != "what"})
    ex = result.get("exact")
    if isinstance(ex, dict):
        add("exact", {k: v for k, v in ex.items() if k not in ("kernel", "kernel_only_evidence")})
    ha = result.get("host_api")
    if isinstance(ha, dict):
        add("host_api", {k: v for k, v in ha.items() if k != "what"})
    if result.get("build_roofline"):
        br = result["build_roofline"]
        add("build", {"rows_per_s": result.get("build_rows_per_s"), "build_s": result.get("build_s"),
                      "kernel_ms": result.get("build_kernel_ms"), "phase_a_gbs": br.get("achieved"),
                      "phase_a_frac": (br.get("achieved") or 0.0) / HBM_PEAK_GBS, "distances_per_row": br.get("distances_per_row"),
                      "link_repair_distances_per_row": br.get("link_repair_distances_per_row")})
    for step in result.get("crud") or []:
        add("crud", step)
    for leg in result.get("legs") or []:
        add("leg", {k: leg.get(k) for k in ("function", "operand", "ms_per_launch", "gbs", "frac")})
    rp = result.get("repeat_detail")
    if isinstance(rp, dict):
        add("repeat", {k: v for k, v in rp.items() if k != "what"})
    ac = (result.get("cpu_baseline") or {}).get("all_cores")
    if isinstance(ac, dict):
        add("cpu_all_cores", {k: v for k, v in ac.items() if k != "note"})
    return lines


def default_sidecar(config_id):
    return os.path.join(ROOT, "gpurun_out", "bench_full_%s.json" % config_id)


def emit(result, sidecar=None, extras=()):
    """Print a result the way the contract above says and return the compact last line.  `extras` = [(name, object)]."""
    side = sidecar or default_sidecar(result.get("config_id", "c3"))
    try:
        os.makedirs(os.path.dirname(side), exist_ok=True)
        with open(side, "w") as f:
            json.dump(result, f, indent=1)
    except OSError as e:  # a read-only tree must not cost the line
        sys.stderr.write("bench.py: sidecar %s not written: %r\n" % (side, e))
    for line in detail_lines(result):
        print(json.dumps(line))
    for name, obj in extras:
        print(json.dumps(extra_line(name, obj)))
    if extras and isinstance(result.get("repeat_detail"), dict):  # (once more next to the headline: the driver keeps a tail)
        print(json.dumps(rounded({"detail": "repeat", **{k: v for k, v in result["repeat_detail"].items() if k != "what"}}, 5)))
    last = compact_line(result)
    print(json.dumps(last))
    sys.stdout.flush()
    return last


def finish(result, sidecar=None):
    """Print the lines; a run whose reference answers disagree with the engine's is a FAILED run (exit code 4)."""
    emit(result, sidecar)
    a = (result.get("cpu_baseline") or {}).get("agreement")
    if a is not None and not agreement_ok(a):
        sys.stderr.write("bench.py: reference agreement below the bar: %s\n" % json.dumps(a))
        sys.exit(4)


def main_c5(args):
    """One shard of BASELINE.json configs[4] (100M rows FLOAT[1536] ip top-100 over 8 GPUs = 12.5M rows per GPU): bulk build,
    ef_search swept to recall@100 >= the target against the exact path, batched search, delete 1 %, insert 1 %, vss_compact,
    recall@100 after every step.  A step is one 1024-query batch; `value` is measured on the freshly built shard."""
    assert int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1, "--config c5 measures ONE shard on one GPU"
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    pkg = load_package()
    rows = args.rows if args.rows != 10_000_000 else 12_500_000
    dim = args.dim if args.dim != 768 else 1536
    metric, k, B, M, efc = "ip", 100, args.batch, 32, 128
    M0, extra = 2 * M, rows // 100
    gen = Mixture(rows + extra, dim, True, device)
    index = pkg.GpuIndex(dim, metric, M, M0, efc, 256)
    index.reserve(rows + extra)

    def stage(first, n, key0, chunk_shift=0):
        for c in range(0, n, CHUNK):
            m = min(CHUNK, n - c)
            x = gen.rows(DATA_SEED, (first + c) // CHUNK + chunk_shift, m)
            ids = torch.arange(key0 + c, key0 + c + m, dtype=torch.int64, device=device)
            torch.cuda.synchronize()
            index.stage_device(ids.data_ptr(), x.data_ptr(), m)
            del x, ids

    stage(0, rows, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    index.build_finalize()
    t_build = time.perf_counter() - t0
    G = max(1, min(32, args.coalesce))
    Q = [gen.rows(QUERY_SEED, i, B) for i in range(G)]
    outs = [(torch.empty((B, k), dtype=torch.int64, device=device), torch.empty((B, k), dtype=torch.float32, device=device),
             torch.empty(B, dtype=torch.int32, device=device)) for _ in range(G)]
    truth = torch.empty((B, k), dtype=torch.int64, device=device)
    state = {"ef": args.ef or 256}

    def probe(n_launches):
        """n_launches launches of G batches each, one at a time; returns (seconds, kernel ms, distances, expansions)."""
        kms, nd, ne = 0.0, 0, 0
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_launches):
            index.search_multi_begin(0, [q.data_ptr() for q in Q], B, k, state["ef"], [o[0].data_ptr() for o in outs],
                                     [o[1].data_ptr() for o in outs], [o[2].data_ptr() for o in outs])
            index.search_end(0)
            kms += index.timing()["search_kernel_ms"]
            st = index.last_search_stats()
            nd, ne = nd + int(st[0]), ne + int(st[1])
        torch.cuda.synchronize()
        return time.perf_counter() - t1, kms, nd, ne

    def exact_now():
        index.search_batch_device(Q[0].data_ptr(), B, k, 0, truth.data_ptr(), outs[0][1].data_ptr(), outs[0][2].data_ptr(), exact=True)
        torch.cuda.synchronize()

    def recall_now(fresh_truth=True):
        if fresh_truth:
            exact_now()
        index.search_batch_device(Q[0].data_ptr(), B, k, state["ef"], outs[0][0].data_ptr(), outs[0][1].data_ptr(), outs[0][2].data_ptr())
        torch.cuda.synchronize()
        return recall_at_k(outs[0][0], truth), outs[0][0].cpu().numpy()

    # ef_search: bench.select_ef's rule on the selection batch (the smallest ef of the sweep whose recall@100 clears the target
    # by two standard errors; register lists hold up to 512 entries); the REPORTED recall is measured on two held-out batches
    exact_now()

    def recalls_at(e):
        state["ef"] = e
        index.search_batch_device(Q[0].data_ptr(), B, k, e, outs[0][0].data_ptr(), outs[0][1].data_ptr(), outs[0][2].data_ptr())
        torch.cuda.synchronize()
        return recall_per_query(outs[0][0], truth)

    ef_pick, sel_recall, sel_se, sweep_log = select_ef(recalls_at, [args.ef] if args.ef else [128, 160, 192, 224, 256, 288, 320, 352, 384,
                                                                                             416, 448, 480, 512], args.target_recall)
    state["ef"] = ef_pick
    held = []
    held_truth = torch.empty((B, k), dtype=torch.int64, device=device)
    for i in range(2):
        qh_ = gen.rows(QUERY_SEED, 5000 + i, B)
        torch.cuda.synchronize()
        index.search_batch_device(qh_.data_ptr(), B, k, 0, held_truth.data_ptr(), outs[1][1].data_ptr(), outs[1][2].data_ptr(), exact=True)
        index.search_batch_device(qh_.data_ptr(), B, k, ef_pick, outs[1][0].data_ptr(), outs[1][1].data_ptr(), outs[1][2].data_ptr())
        torch.cuda.synchronize()
        held += recall_per_query(outs[1][0], held_truth)
        del qh_
    recall, recall_se = mean_and_se(held)
    ef = state["ef"]
    n_launches = max(1, (max(1, args.steps) + G - 1) // G)
    probe(max(1, (args.warmup + G - 1) // G))
    elapsed, kernel_ms, dists, expans = probe(n_launches)
    steps = n_launches * G
    bytes_per_launch = (dists * (4 * dim + 4) + expans * (4 + 4 * M0)) / n_launches
    # the CRUD part of the configuration, each step followed by recall against a fresh exact answer and one timed launch
    crud = []
    g = torch.Generator(device="cpu").manual_seed(1234)
    dead = torch.randperm(rows, generator=g)[:extra].numpy().astype(np.int64)
    t1 = time.perf_counter()
    removed = index.remove(dead)
    t_del = time.perf_counter() - t1
    r, got = recall_now()
    dt, ms, _, _ = probe(1)
    crud.append({"after": "deleting 1 %% (%d rows, %.2f s)" % (removed, t_del), "recall_at_100": round(r, 4),
                 "queries_per_s": G * B / dt, "deleted_rows_returned": int(np.isin(got, dead).sum())})
    t1 = time.perf_counter()
    stage(0, extra, rows, chunk_shift=100_000)
    index.build_finalize()
    t_ins = time.perf_counter() - t1
    r, got = recall_now()
    dt, ms, _, _ = probe(1)
    crud.append({"after": "inserting 1 %% (%.2f s)" % t_ins, "recall_at_100": round(r, 4), "queries_per_s": G * B / dt,
                 "deleted_rows_returned": int(np.isin(got, dead).sum())})
    t1 = time.perf_counter()
    index.compact()
    t_compact = time.perf_counter() - t1
    r, got = recall_now()
    dt, ms, _, _ = probe(1)
    crud.append({"after": "vss_compact (%.2f s)" % t_compact, "recall_at_100": round(r, 4), "queries_per_s": G * B / dt,
                 "deleted_rows_returned": int(np.isin(got, dead).sum()), "nodes": int(index.nodes()), "size": int(index.size())})
    full = (rows, dim) == (12_500_000, 1536)
    result = {
        "metric": "queries/sec at recall@100, one 12.5M-row shard of 100M×1536 FLOAT ip top-100 (BASELINE configs[4]); index "
                  "build rows/sec",
        "config_id": "c5",
        "value": steps * B / elapsed, "unit": "queries/s", "n_gpus": 1, "steps": steps, "warmup": args.warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "recall_at_100": round(recall, 4), "recall_at_100_se": round(recall_se, 5),
        "recall": {"reported": "held-out", "heldout": {"batches": 2, "queries": len(held), "mean": round(recall, 5), "se": round(recall_se, 5)},
                   "selection": {"batches": 1, "queries": B, "mean": round(sel_recall, 5), "se": round(sel_se, 5),
                                 "rule": "smallest ef of the sweep with mean - 2 se >= target"}},
        "ef_search": ef, "ef_sweep": sweep_log,
        "build_rows_per_s": rows / t_build, "build_s": t_build,
        "crud": crud, "compact_s": t_compact,
        "config": {"workload": ("one shard (12.5M rows = 1/8) of configs[4]: 100M rows FLOAT[1536] ip top-100, batched 1024 queries, "
                                "then delete 1 % / insert 1 % / compact" if full else
                                "DEVELOPMENT RUN (not the benchmark): %d rows FLOAT[%d] ip top-100 shard" % (rows, dim)),
                   "rows": rows, "dim": dim, "index_metric": metric, "k": k, "batch_queries": B, "M": M, "M0": M0,
                   "ef_construction": efc, "ef_search": ef, "batches_per_launch": G, "launches_in_flight": 1,
                   "parallelism": "one shard of shard8"},
        "roofline": {"bound": "hbm", "kernel": "k_search", "achieved": bytes_per_launch / (kernel_ms / n_launches / 1e3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": bytes_per_launch / (kernel_ms / n_launches / 1e3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": bytes_per_launch, "avg_kernel_ms": kernel_ms / n_launches,
                     "launches": n_launches, "distances_per_query": dists / steps / B, "expansions_per_query": expans / steps / B,
                     "visited_set": visited_set_form(max(k, ef), rows)},
        "cpu_baseline": None,
    }
    # HBM traffic of k_search from the committed rocprofv3 --pmc passes of this configuration (profiles/); attached only when the
    # configuration AND the launch shape (batches per launch) match
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_k_search_config4_shard_full_size.json"))):
            pm = json.load(open(path))
            c = pm["config"]
            if (c["rows"], c["dim"], c["index_metric"], c["M"], c["M0"], c["ef_construction"], c["ef_search"], c["batch_queries"], c["k"]) == \
                    (rows, dim, metric, M, M0, efc, ef, B, k):
                g_pm = pm.get("batches_per_launch", 16)
                if g_pm != G:  # (a figure counted on another launch shape is not this run's — not scaled, not attached)
                    result["roofline"]["traffic_source"] = "%s was counted on launches of %d batches, this run carries %d: not attached" % (
                        os.path.relpath(path, ROOT), g_pm, G)
                    continue
                result["roofline"]["traffic"] = pm["hbm_bytes_per_launch"]
                result["roofline"]["traffic_over_algorithmic"] = result["roofline"]["traffic"] / bytes_per_launch
                result["roofline"]["traffic_source"] = os.path.relpath(path, ROOT) + " (committed rocprofv3 --pmc passes, not this run's counters)"
    except Exception:  # noqa: BLE001
        pass
    if not args.no_cpu_baseline:
        # the reference library, one thread, on a PREFIX index of the same data (the shard itself is an 80 GB stream): graph
        # built by the engine with identical options, handed over through the reference stream format, same queries, same ef
        from oracle_lib import CpuIndex, load_oracle, load_ref
        lib, kind = load_ref(), "reference"
        if lib is None:
            lib, kind = load_oracle(), "port"
        n_pre = min(rows, args.cpu_sample_rows if args.cpu_sample_rows != 200_000 else 1_000_000)
        index.close()
        del index
        x = gen.rows(DATA_SEED, 0, n_pre)
        ids = torch.arange(n_pre, dtype=torch.int64, device=device)
        pre = pkg.GpuIndex(dim, metric, M, M0, efc, ef)
        pre.reserve(n_pre)
        pre.stage_device(ids.data_ptr(), x.data_ptr(), n_pre)
        pre.build_finalize()
        cpu = CpuIndex(lib, dim, metric, M, M0, efc, ef)
        cpu.load(pre.save())
        qh = Q[0].cpu().numpy()
        gk, gd, _ = pre.search_batch(qh, k, ef)
        for i in range(8):
            cpu.search(qh[i], k, ef=ef)
        ck, cd = np.full((B, k), -1, dtype=np.int64), np.full((B, k), np.inf, dtype=np.float32)
        t1, n = time.perf_counter(), 0
        while time.perf_counter() - t1 < args.cpu_seconds:
            kk, dd, _ = cpu.search(qh[n % B], k, ef=ef)
            if n < B:
                ck[n, :len(kk)], cd[n, :len(dd)] = kk, dd
            n += 1
        dt = time.perf_counter() - t1
        result["cpu_baseline"] = {
            "value": n / dt, "unit": "queries/s", "cores": 1, "kind": kind, "index_rows": n_pre, "cpu_model": cpu_model_name(),
            "sample": "%d single-thread ef_search(k=%d, ef=%d) calls on a %d-row PREFIX index of the same data (graph built by the "
                      "engine with identical options, handed over through the reference stream format; a graph 12.5x smaller than "
                      "the shard favours the CPU), same queries" % (n, k, ef, n_pre),
            "agreement": agreement(ck[:min(n, B)], cd[:min(n, B)], gk, gd, metric, ef, queries=qh[:min(n, B)],
                                   fetch_rows=lambda keys: {int(key): x[int(key)].cpu().numpy() for key in keys},
                                   ref_distance=make_ref_distance(lib, metric, dim),
                                   replay=make_wave_order_replay(lambda: [stream_of(pre)], dim, metric, M, M0, efc, k, ef, qh))}
        pre.close()
    finish(result, args.sidecar)


def main_c2(args):
    """BASELINE.json configs[1]: 1M rows FLOAT[128] l2sq top-10, single MI355X, single-query HNSW_INDEX_SCAN."""
    assert int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1, "configs[1] is a single-GPU configuration"
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    pkg = load_package()
    rows = args.rows if args.rows != 10_000_000 else 1_000_000
    dim = args.dim if args.dim != 768 else 128
    metric, k, M, M0, efc, ef = "l2sq", args.k, 16, 32, 128, args.ef or 64  # the reference's default index options
    gen = Mixture(rows, dim, False, device)
    index = pkg.GpuIndex(dim, metric, M, M0, efc, ef)
    index.reserve(rows)
    t0 = time.perf_counter()
    for c in range(0, rows, CHUNK):
        m = min(CHUNK, rows - c)
        x = gen.rows(DATA_SEED, c // CHUNK, m)
        ids = torch.arange(c, c + m, dtype=torch.int64, device=device)
        torch.cuda.synchronize()
        index.stage_device(ids.data_ptr(), x.data_ptr(), m)
        del x, ids
    index.build_finalize()
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    steps, warmup = max(1, args.steps), max(args.warmup, 16)
    nq = 4096
    Qd = gen.rows(QUERY_SEED, 0, nq)
    Q = Qd.cpu().numpy()
    truth = index.search_batch(Q[:1024], k, exact=True)[0]
    got = index.search_batch(Q[:1024], k, ef)[0]
    recall = float(np.mean([len(set(got[i].tolist()) & set(truth[i].tolist())) / k for i in range(1024)]))
    st = index.last_search_stats()
    dists_q, exp_q = float(st[0]) / 1024, float(st[1]) / 1024
    for i in range(warmup):
        index.search(Q[i % nq], k, ef)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):  # one HNSW_INDEX_SCAN probe per step
        index.search(Q[i % nq], k, ef)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kms, nk = 0.0, 512  # kernel time of the same calls (hipEvents on the probe's stream), outside the timed region
    index.set_search_probe_wait(False)  # wait on the stream, with events around the kernel
    for i in range(nk):
        index.search(Q[i % nq], k, ef)
        kms += index.timing()["search_kernel_ms"]
    kernel_us = kms / nk * 1e3
    t0 = time.perf_counter()
    for i in range(nk):
        index.search(Q[i % nq], k, ef)
    stream_wait_us = (time.perf_counter() - t0) / nk * 1e6
    index.set_search_probe_wait(True)
    # one chunk of HNSW_INDEX_JOIN: floor(2048 / k) queries per Execute call (reference hnsw_optimize_join.cpp:111-168, which
    # answers them one ef_search after the other on one thread); here one vss_search_batch call, host pointers both ways
    chunk = max(1, 2048 // k)
    for i in range(4):
        index.search_batch(Q[i * chunk:(i + 1) * chunk], k, ef)
    n_chunks = min(16, nq // chunk)
    t0 = time.perf_counter()
    for i in range(n_chunks):
        index.search_batch(Q[i * chunk:(i + 1) * chunk], k, ef)
    join_chunk_us = (time.perf_counter() - t0) / n_chunks * 1e6
    bytes_q = dists_q * (4 * dim + 4) + exp_q * (4 + 4 * M0)
    result = {
        "metric": "queries/sec, single-query HNSW_INDEX_SCAN, 1M×128 FLOAT l2sq top-10 (BASELINE configs[1]); index build "
                  "rows/sec",
        "config_id": "c2",
        "value": steps / elapsed, "unit": "queries/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "recall_at_10": round(recall, 4), "ef_search": ef, "build_rows_per_s": rows / t_build,
        "join_chunk": {"queries": chunk, "us_per_chunk": join_chunk_us, "queries_per_s": chunk / (join_chunk_us * 1e-6),
                       "what": "one vss_search_batch call of floor(2048 / k) queries, host pointers (the HNSW_INDEX_JOIN chunk)"},
        "config": {"workload": "configs[1]: 1M rows FLOAT[128] l2sq top-10, single MI355X, single-query HNSW_INDEX_SCAN "
                               "(one vss_search call per step)" if (rows, dim) == (1_000_000, 128) else
                               "DEVELOPMENT RUN (not the benchmark): %d rows FLOAT[%d] l2sq single-query" % (rows, dim),
                   "rows": rows, "dim": dim, "index_metric": metric, "k": k, "batch_queries": 1, "M": M, "M0": M0,
                   "ef_construction": efc, "ef_search": ef, "parallelism": "single"},
        "roofline": {"bound": "hbm", "kernel": "k_search_solo (one walking wave + 7 helper waves per query)",
                     "achieved": bytes_q / (kernel_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": bytes_q / (kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": bytes_q, "avg_kernel_ms": kernel_us / 1e3,
                     "latency_bound": True, "us_per_expansion": kernel_us / max(exp_q, 1.0), "expansions_per_query": exp_q,
                     "us_per_call_waiting_on_the_stream": stream_wait_us,
                     "distances_per_query": dists_q,
                     "note": "one query = a chain of dependent expansions (neighbour list, then the rows it names): the kernel "
                             "is bound by HBM round-trip latency, not bandwidth"},
    }
    if not args.no_cpu_baseline:
        from oracle_lib import CpuIndex, load_oracle, load_ref
        lib, kind = load_ref(), "reference"
        if lib is None:
            lib, kind = load_oracle(), "port"
        cpu = CpuIndex(lib, dim, metric, M, M0, efc, ef)
        cpu.load(index.save())
        for i in range(64):
            cpu.search(Q[i], k, ef=ef)
        n_keep = 2048
        ck, cd = np.full((n_keep, k), -1, dtype=np.int64), np.full((n_keep, k), np.inf, dtype=np.float32)
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < args.cpu_seconds:
            kk, dd, _ = cpu.search(Q[n % nq], k, ef=ef)
            if n < n_keep:
                ck[n, :len(kk)], cd[n, :len(dd)] = kk, dd
            n += 1
        dt = time.perf_counter() - t0
        gk, gd, _ = index.search_batch(Q[:n_keep], k, ef)
        result["join_chunk"]["reference_us_per_chunk"] = chunk / (n / dt) * 1e6  # the same thread, one ef_search after the other
        result["cpu_baseline"] = {"value": n / dt, "unit": "queries/s", "cores": 1, "kind": kind, "cpu_model": cpu_model_name(),
                                  "sample": "%d single-thread ef_search(k=%d, ef=%d) calls on the same %d-row graph (built by the "
                                            "engine, handed over through the reference stream format), same queries" % (n, k, ef, rows),
                                  "agreement": agreement(ck[:min(n, n_keep)], cd[:min(n, n_keep)], gk, gd, metric, ef,
                                                         queries=Q[:min(n, n_keep)],
                                                         fetch_rows=make_row_fetch(gen, rows, full_chunks=False),
                                                         ref_distance=make_ref_distance(lib, metric, dim),
                                                         replay=make_wave_order_replay(lambda: [stream_of(index)], dim, metric, M,
                                                                                       M0, efc, k, ef, Q))}
    finish(result, args.sidecar)


def main_a13(args):
    """array_distance / array_cosine_distance / array_negative_inner_product over a resident FLOAT[768]
    column (the brute-force plan's projection, and the k fetched rows after an index scan; names at reference
    hnsw_index.cpp:659-673) — one streaming pass, HBM-bound: 4 * dim bytes per row in (twice that with a column operand),
    4 bytes out.  PARITY UNPINNED (DuckDB v1.4.3's core source is not in the reference tree): checked against the README
    values and an fp64 formula in tests/, never against DuckDB itself."""
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    pkg = load_package()
    lib = pkg.load_library()
    rows = min(args.rows, 4_000_000) if args.rows != 10_000_000 else 4_000_000
    dim = args.dim
    g = torch.Generator(device=device).manual_seed(DATA_SEED)
    a = torch.randn(rows, dim, generator=g, device=device)
    b = torch.randn(rows, dim, generator=g, device=device)
    out = torch.empty(rows, dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    legs = []
    for fn, name in enumerate(("array_distance", "array_cosine_distance", "array_negative_inner_product")):
        for b_const in (1, 0):
            reps = max(1, args.steps if args.steps else 20)
            for _ in range(3):
                assert lib.vss_distance_batch_device(fn, a.data_ptr(), b.data_ptr(), b_const, rows, dim, out.data_ptr(), stream) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                lib.vss_distance_batch_device(fn, a.data_ptr(), b.data_ptr(), b_const, rows, dim, out.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            nbytes = 4.0 * dim * rows * (1 if b_const else 2) + 4.0 * rows
            legs.append({"function": name, "operand": "constant" if b_const else "column", "ms_per_launch": ms,
                         "rows_per_s": rows / (ms / 1e3), "algorithmic_bytes_per_launch": nbytes,
                         "gbs": nbytes / (ms / 1e3) / 1e9, "frac": nbytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS})
    # spot check of the arithmetic on this very data (fp64 formula; the collected tests hold the full comparison)
    x, y = a[:2048].double(), b[:2048].double()
    lib.vss_distance_batch_device(0, a.data_ptr(), b.data_ptr(), 0, 2048, dim, out.data_ptr(), stream)
    torch.cuda.synchronize()
    err = float(((out[:2048].double() - (x - y).norm(dim=1)).abs() / (x - y).norm(dim=1)).max())
    worst = min(legs, key=lambda l: l["frac"])
    # HBM traffic of the slowest leg from the committed rocprofv3 --pmc FETCH_SIZE pass of this same command (reads only: the
    # 4 bytes per row written are 0.1 % of the launch and were not counted), attached when the pass was taken on this shape
    traffic, traffic_src = None, None
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*a13*rocprof*.json"))):
            pm = json.load(open(path))
            if (pm["line"]["config"]["rows"], pm["line"]["config"]["dim"]) != (rows, dim):
                continue
            for leg in pm.get("legs_with_traffic", []):
                if (leg["function"], leg["operand"]) == (worst["function"], worst["operand"]) and leg.get("hbm_read_bytes"):
                    traffic, traffic_src = float(leg["hbm_read_bytes"]), os.path.relpath(path, ROOT) + " (FETCH_SIZE x 2, reads only)"
    except Exception:  # noqa: BLE001
        pass
    chunk = a13_host_chunks(pkg, lib, args)
    result = {
        "metric": "rows/sec, array_distance / array_cosine_distance / array_negative_inner_product over a resident FLOAT[%d] "
                  "column (SURVEY §8 row a13)" % dim,
        "config_id": "a13", "value": worst["rows_per_s"], "unit": "rows/s", "n_gpus": 1, "steps": max(1, args.steps or 20),
        "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%d rows FLOAT[%d], the three array_* functions, constant and column operand" % (rows, dim),
                   "rows": rows, "dim": dim},
        "roofline": {"bound": "hbm", "kernel": "k_array_distance", "achieved": worst["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": worst["frac"], "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_over_algorithmic": traffic / worst["algorithmic_bytes_per_launch"] if traffic else None,
                     "algorithmic_bytes_per_launch": worst["algorithmic_bytes_per_launch"],
                     "avg_kernel_ms": worst["ms_per_launch"], "slowest_leg": "%s, %s operand" % (worst["function"], worst["operand"]),
                     "timed_with": "events on the stream the kernel is launched on, %d back-to-back launches per leg" % max(1, args.steps or 20)},
        "legs": legs, "spot_check_max_rel_err_vs_fp64": err, "host_chunks": chunk,
        "chunk_2048_us": chunk["chunk_2048_us"], "cpu_chunk_2048_us": chunk["cpu_chunk_2048_us"],
        "crossover_rows": chunk["crossover_rows"],
        "parity": "UNPINNED: DuckDB v1.4.3 core source absent from the reference tree (SURVEY §8c); README values + fp64 formula only",
        "cpu_baseline": None,
    }
    finish(result, args.sidecar)


def a13_host_chunks(pkg, lib, args):
    """Row a13 at the call shape DuckDB actually gives a scalar function: one <= 2048-row chunk of HOST-resident
    FLOAT[N] per call, the other operand a constant — `vss_distance_batch` (H2D of the chunk, kernel, D2H of 2048 floats; PCIe
    inclusive) beside ONE host thread running the sequential f32 loop SURVEY Appendix B states for DuckDB's functions
    (oracle/hnsw_oracle.cpp orc_array_function, kind "port": DuckDB's own source is not in the reference tree), on the same
    chunks, and the row count per call from which the device wins."""
    from oracle_lib import load_oracle
    orc = load_oracle()
    rng = np.random.default_rng(DATA_SEED)
    legs, first_win = [], {}
    for dim in (768, 1536):
        top = 131072
        base = rng.random((top + 6 * 2048, dim), dtype=np.float32) - 0.5  # (one pool per dimension; the chunks are windows of it)
        for rows in (2048, 8192, 32768, top):
            bufs = [base[i * 2048:i * 2048 + rows] for i in range(6)]
            q = rng.standard_normal(dim, dtype=np.float32)
            out_g, out_c = np.empty(rows, dtype=np.float32), np.empty(rows, dtype=np.float32)
            for fn, name in ((0, "array_distance"), (1, "array_cosine_distance")):
                reps = max(3, min(200, (1 << 22) // rows))
                for i in range(3):
                    assert lib.vss_distance_batch(fn, bufs[i % len(bufs)].ctypes.data, q.ctypes.data, 1, rows, dim,
                                                  out_g.ctypes.data, 0) == 0
                t0 = time.perf_counter()
                for i in range(reps):
                    lib.vss_distance_batch(fn, bufs[i % len(bufs)].ctypes.data, q.ctypes.data, 1, rows, dim, out_g.ctypes.data, 0)
                gpu_us = (time.perf_counter() - t0) / reps * 1e6
                creps = max(2, min(50, (1 << 20) // rows))
                orc.orc_array_function(fn, bufs[0].ctypes.data, q.ctypes.data, 1, rows, dim, out_c.ctypes.data)
                t0 = time.perf_counter()
                for i in range(creps):
                    orc.orc_array_function(fn, bufs[i % len(bufs)].ctypes.data, q.ctypes.data, 1, rows, dim, out_c.ctypes.data)
                cpu_us = (time.perf_counter() - t0) / creps * 1e6
                lib.vss_distance_batch(fn, bufs[(creps - 1) % len(bufs)].ctypes.data, q.ctypes.data, 1, rows, dim, out_g.ctypes.data, 0)
                err = float(np.max(np.abs(out_g - out_c) / np.maximum(1e-6, np.abs(out_c))))
                legs.append({"function": name, "dim": dim, "rows_per_call": rows, "device_us": gpu_us, "cpu_thread_us": cpu_us,
                             "device_GBs_incl_pcie": rows * dim * 4 / gpu_us / 1e3, "max_rel_diff": err})
                if gpu_us < cpu_us and (name, dim) not in first_win:
                    first_win[(name, dim)] = rows
            del bufs
        del base

    def at(name, dim, rows, key):
        return next(l[key] for l in legs if (l["function"], l["dim"], l["rows_per_call"]) == (name, dim, rows))
    return {"legs": legs,
            "chunk_2048_us": {"768": at("array_distance", 768, 2048, "device_us"), "1536": at("array_distance", 1536, 2048, "device_us")},
            "cpu_chunk_2048_us": {"768": at("array_distance", 768, 2048, "cpu_thread_us"),
                                  "1536": at("array_distance", 1536, 2048, "cpu_thread_us")},
            "crossover_rows": {"%s/%d" % (key[0].replace("array_", "").replace("_distance", ""), key[1]): first_win.get(key)
                               for key in (("array_distance", 768), ("array_distance", 1536), ("array_cosine_distance", 768),
                                           ("array_cosine_distance", 1536))},
            "cpu": {"kind": "port", "threads": 1, "cpu_model": cpu_model_name(),
                    "what": "sequential f32 loop per row (SURVEY Appendix B), g++ -O3, no fast-math: PARITY UNPINNED"},
            "what": "vss_distance_batch on pageable host chunks, constant second operand; rows per call swept for the crossover "
                    "(null = the host thread was faster at every size tried)"}


QUALITY_OPTIONS = [(16, 128), (32, 384)]  # (M, ef_construction): the reference defaults and the headline's options
QUALITY_EFS = [32, 64, 128, 256, 512, 1024]


def effective_cpus():
    """Host threads worth starting: the affinity mask, capped by the cgroup's CPU quota (the GPU box shows 256 CPUs under a quota
    of 16 — more add() streams than that only spin on each other's node locks)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def quality_study(pkg, x_dev, Q_dev, metric, options, efs, k, threads, device_index=0, log=None):
    """Build-quality parity: the batch-synchronous GPU build inserts up to rows/32 (at most 32 768) nodes per
    batch that do not see each other, where the reference runs N interleaved add() streams over one shared graph
    (hnsw_index_physical_create.cpp:148-209, 235-247).  For every (M, ef_construction): graph A = the REFERENCE LIBRARY building
    the rows on `threads` host threads (oracle/_ref: its own add(), one stream per thread over 2048-row chunks — what CREATE INDEX
    runs), handed to the engine through the stream format (vss_load); graph B = the engine's bulk build of the same rows.  BOTH
    are searched by the engine, same queries, same ef grid, recall@k against the exact path over the same rows.  Returns one
    object per option pair."""
    from oracle_lib import CpuIndex, load_ref
    ref = load_ref()
    if ref is None:
        raise RuntimeError("oracle/_ref/libusearch_ref.so is not here: the reference build cannot be run")
    rows, dim = x_dev.shape
    nq = Q_dev.shape[0]
    device = x_dev.device
    x_host = x_dev.cpu().numpy()
    ids = torch.arange(rows, dtype=torch.int64, device=device)
    ok = torch.empty((nq, k), dtype=torch.int64, device=device)
    od = torch.empty((nq, k), dtype=torch.float32, device=device)
    oc = torch.empty(nq, dtype=torch.int32, device=device)
    truth = None
    out = []
    for M, efc in options:
        eb = pkg.GpuIndex(dim, metric, M, 2 * M, efc, 64, device=device_index)
        eb.reserve(rows)
        if os.environ.get("VSS_QUALITY_BUILD_PARAMS"):  # experiments: "max_batch,growth_div" instead of the engine's 32768,32
            eb.set_build_params(*[int(v) for v in os.environ["VSS_QUALITY_BUILD_PARAMS"].split(",")])
        torch.cuda.synchronize()
        eb.stage_device(ids.data_ptr(), x_dev.data_ptr(), rows)
        t0 = time.perf_counter()
        eb.build_finalize()
        t_engine = time.perf_counter() - t0
        batches = eb.timing(reset=True)["build_batches"]
        if truth is None:
            eb.search_batch_device(Q_dev.data_ptr(), nq, k, 0, ok.data_ptr(), od.data_ptr(), oc.data_ptr(), exact=True)
            torch.cuda.synchronize()
            truth = ok.clone()
        cpu = CpuIndex(ref, dim, metric, M, 2 * M, efc, 64)
        t_ref, n_ref = cpu.add_mt(np.arange(rows, dtype=np.int64), x_host, threads)
        assert n_ref == rows
        blob = cpu.save()
        del cpu
        ea = pkg.GpuIndex(dim, metric, M, 2 * M, efc, 64, device=device_index)
        ea.load(blob)
        del blob
        assert ea.size() == rows
        per_ef = []
        for ef in efs:
            row = {"ef": ef}
            for name, ix in (("A", ea), ("B", eb)):
                ix.search_batch_device(Q_dev.data_ptr(), nq, k, ef, ok.data_ptr(), od.data_ptr(), oc.data_ptr())
                torch.cuda.synchronize()
                st = ix.last_search_stats()
                row[name] = round(recall_at_k(ok, truth), 4)
                row["distances_" + name] = round(float(st[0]) / nq, 1)
            row["B_minus_A"] = round(row["B"] - row["A"], 4)
            per_ef.append(row)
        stats = {}
        for name, ix in (("A", ea), ("B", eb)):  # level-0 density: directed links per node
            ls = ix.level_stats(0)
            stats["links0_per_node_" + name] = round(float(ls[1]) / max(1, float(ls[0])), 2)
        o = {"M": M, "M0": 2 * M, "ef_construction": efc, "rows": rows, "dim": dim, "metric": metric, "queries": nq, "k": k,
             "reference_build": {"threads": threads, "seconds": round(t_ref, 2), "rows_per_s": round(rows / t_ref, 1)},
             "engine_build": {"seconds": round(t_engine, 3), "rows_per_s": round(rows / t_engine, 1), "batches": batches},
             "per_ef": per_ef, "max_abs_B_minus_A": max(abs(r["B_minus_A"]) for r in per_ef),
             "min_B_minus_A": min(r["B_minus_A"] for r in per_ef), **stats}
        out.append(o)
        if log:
            log(o)
        ea.close()
        eb.close()
    return out


def main_quality(args):
    """`bench.py --config quality` — see quality_study.  Rows = a prefix of the benchmark's own mixture (the first
    --quality-rows rows of chunk 0 of the 10M x 768 cosine data), queries from the benchmark's query stream."""
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    pkg = load_package()
    dim, k = args.dim, args.k
    metric = args.metric or "cosine"
    rows = args.quality_rows
    gen = Mixture(args.rows, dim, metric != "l2sq", device)
    x = torch.cat([gen.rows(DATA_SEED, c, CHUNK) for c in range((rows + CHUNK - 1) // CHUNK)])[:rows].contiguous()
    Q = torch.cat([gen.rows(QUERY_SEED, i, 1024) for i in range(2)]).contiguous()
    torch.cuda.synchronize()
    threads = effective_cpus()
    t0 = time.perf_counter()
    options = QUALITY_OPTIONS if not args.quality_options else \
        [tuple(int(v) for v in item.split("/")) for item in args.quality_options.split(",")]
    efs = QUALITY_EFS if not args.quality_efs else [int(v) for v in args.quality_efs.split(",")]
    study = quality_study(pkg, x, Q, metric, options, efs, k, threads,
                          log=lambda o: sys.stderr.write("quality: %s\n" % json.dumps(o)))
    worst = max(o["max_abs_B_minus_A"] for o in study)
    lowest = min(o["min_B_minus_A"] for o in study)
    default = study[0]
    finish({
        "metric": "recall@%d of the engine-built graph (B) minus recall@%d of a reference-built graph (A), same rows, same "
                  "options, both searched by the engine" % (k, k),
        "config_id": "quality", "value": lowest, "unit": "recall B - A, lowest over the ef grid",
        "n_gpus": 1, "steps": len(efs) * len(options), "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%d-row prefix of the benchmark's 10M x %d %s mixture; graphs at (M, ef_construction) = %s; %d queries; "
                               "ef_search grid %s" % (rows, dim, metric, options, Q.shape[0], efs),
                   "rows": rows, "dim": dim, "index_metric": metric, "k": k},
        "quality": study, "max_abs_B_minus_A": worst, "within_0.01": bool(worst <= 0.01),
        "quality_compact": {"%d/%d" % (o["M"], o["ef_construction"]): [[r["ef"], r["A"], r["B"]] for r in o["per_ef"]] for o in study},
        # the reference-default options at the largest ef of the grid: what "the index cannot reach the target" means for BOTH builds
        "plateau": {"M": default["M"], "ef_construction": default["ef_construction"], "rows": rows, "at_ef_search": efs[-1],
                    "engine": default["per_ef"][-1]["B"], "reference": default["per_ef"][-1]["A"]},
        "roofline": None, "cpu_baseline": None, "study_wall_s": round(time.perf_counter() - t0, 1),
    }, args.sidecar)


def small_launches(index, gen, k, ef, B_join):
    """The reference's two real SQL surfaces on the headline index, as latencies: ONE query per call (HNSW_INDEX_SCAN,
    reference hnsw_index_scan.cpp:43-90 -> vss_search) and one chunk of HNSW_INDEX_JOIN (floor(2048 / k) queries per Execute,
    hnsw_optimize_join.cpp:111-168 -> one vss_search_batch call, host pointers both ways).  Every call sees queries no
    earlier call has seen."""
    qh = torch.cat([gen.rows(QUERY_SEED, 7000 + i, 1024) for i in range(4)]).cpu().numpy()
    for i in range(24):
        index.search(qh[i], k, ef)
    n1 = 400
    t0 = time.perf_counter()
    for i in range(n1):
        index.search(qh[24 + i], k, ef)
    single_us = (time.perf_counter() - t0) / n1 * 1e6
    base = 512
    for i in range(3):
        index.search_batch(qh[base + i * B_join:base + (i + 1) * B_join], k, ef)
    n2 = (len(qh) - base) // B_join - 3
    t0 = time.perf_counter()
    for i in range(3, 3 + n2):
        index.search_batch(qh[base + i * B_join:base + (i + 1) * B_join], k, ef)
    join_us = (time.perf_counter() - t0) / n2 * 1e6
    return {"single_query": {"us_per_call": single_us, "calls": n1, "entry": "vss_search (host pointers, one query)"},
            "join_chunk": {"queries": B_join, "us_per_call": join_us, "calls": n2,
                           "entry": "vss_search_batch (host pointers, floor(2048 / k) queries)"},
            "kernel": "k_search, one walker per workgroup running its scoring waves as a crew (barrier hand-over, DESIGN §4.2d)",
            "ef_search": ef}


EXTRA_CONFIGS = {  # the other BASELINE configurations on the driver's clock: compact forms, one subprocess each
    # the headline workload on an index with the REFERENCE'S DEFAULT options (M 16, M0 32, ef_construction 128: what plain
    # `CREATE INDEX ... USING HNSW` builds) — its recall on this data plateaus below the target, so the line reports the
    # largest ef_search of the sweep; kept beside the headline as the judge of round 3 asked
    "reference_default_options": (["--config", "c3", "--M", "16", "--ef-construction", "128", "--extras", "none", "--steps", "20",
                                   "--warmup", "5", "--no-cpu-baseline", "--regimes", "none", "--host-api-seconds", "0",
                                   "--no-small-launches", "--heldout-batches", "4", "--wide-ef-sweep", "--repeats", "0"], 300),
    # the build half of the metric at ROUND 3's index options (ef_construction 256), so that `build_rows_per_s` stays comparable
    # round over round: the headline index pays 10.5k distances per row for its ef_construction 384, this one 7.1k
    "build_efc256": (["--config", "c3", "--ef-construction", "256", "--build-only", "--extras", "none"], 240),
    "c2": (["--config", "c2", "--steps", "2000", "--cpu-seconds", "6"], 240),
    "c4": (["--config", "c4", "--steps", "40", "--warmup", "10", "--cpu-seconds", "6", "--regimes", "none",
            "--host-api-seconds", "0", "--heldout-batches", "4", "--repeats", "0"], 600),
    "c5": (["--config", "c5", "--steps", "32", "--warmup", "16", "--cpu-seconds", "8"], 600),
    "a13": (["--config", "a13", "--steps", "20"], 180),
    # build-quality parity: the reference library builds a 200k-row prefix on the host's cores, the engine builds the same rows,
    # the engine searches both graphs (LAST: it is the longest, and the first to go when the run's budget is short).  Inside the
    # driver's run: the reference-default options only (a minute of host build); the headline's options (M 32, ef_construction
    # 384: four minutes of host build) and a 1M-row instance are `bench.py --config quality` runs kept under profiles/r06*_quality_*
    "quality": (["--config", "quality", "--quality-options", "16/128"], 300),
}


DEFAULT_EXTRAS = ["c5", "reference_default_options", "c2", "c4", "a13", "build_efc256", "quality"]


def run_extras(which, budget_s, started):
    """Each extra configuration runs in a process of its own (`python bench.py --config ...`), after this one has let go of
    its index: a fault or a time-out there costs that object, never the headline.  Returns {name: its JSON line | error}."""
    import subprocess
    out = {}
    for name in which:
        argv, limit = EXTRA_CONFIGS[name]
        left = budget_s - (time.perf_counter() - started)
        if left < 60:
            out[name] = {"error": "skipped: the run's time budget (%d s) is spent" % budget_s}
            continue
        t0 = time.perf_counter()
        side = default_sidecar("extra_" + name)
        try:
            if os.path.exists(side):
                os.remove(side)
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--sidecar", side] + argv,
                               capture_output=True, text=True, timeout=min(limit, left))
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if lines:
                try:  # the process's complete object; its last stdout line is the compact form of the same
                    out[name] = json.load(open(side))
                except (OSError, ValueError):
                    out[name] = json.loads(lines[-1])
                out[name]["exit_code"] = p.returncode
            else:
                out[name] = {"error": "no JSON line (exit code %d)" % p.returncode, "stderr_tail": p.stderr[-600:]}
        except subprocess.TimeoutExpired:
            out[name] = {"error": "timed out after %.0f s" % (time.perf_counter() - t0)}
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": repr(e)}
        out[name]["wall_s"] = round(time.perf_counter() - t0, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 128 batches; 4000 queries for --config c2)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default: 24; 16 for --config c2)")
    ap.add_argument("--rows", type=int, default=int(os.environ.get("VSS_BENCH_ROWS", 10_000_000)))
    ap.add_argument("--dim", type=int, default=int(os.environ.get("VSS_BENCH_DIM", 768)))
    ap.add_argument("--metric", default=os.environ.get("VSS_BENCH_METRIC", ""))
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--query-batches", type=int, default=8)
    ap.add_argument("--target-recall", type=float, default=0.95)
    ap.add_argument("--heldout-batches", type=int, default=8,
                    help="batches (of --batch queries) the reported recall is measured on; ef_search is selected on two others")
    ap.add_argument("--ef", type=int, default=0, help="fix ef_search instead of sweeping it")
    ap.add_argument("--M", type=int, default=HEADLINE_OPTIONS["M"], help="index option M (reference default 16; see DESIGN.md)")
    ap.add_argument("--M0", type=int, default=0, help="index option M0 (default 2*M as in the reference)")
    ap.add_argument("--ef-construction", type=int, default=HEADLINE_OPTIONS["ef_construction"],
                    help="index option ef_construction (reference default 128; 384 since round 4: fewest bytes per query at recall "
                         "0.95 in the sweep of profiles/r04f_option_sweep_10m768.json)")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("VSS_BENCH_PIPELINE", 3)),
                    help="launches in flight on separate search contexts (the analogue of usearch's per-thread contexts); "
                         "one launch per batch with 1 and 3 in flight is measured after the timed region and reported in "
                         "roofline.regimes")
    ap.add_argument("--coalesce", type=int, default=int(os.environ.get("VSS_BENCH_COALESCE", 32)),
                    help="probe batches answered by one launch of the search engine (vss_search_multi_device_begin); 1 = one "
                         "launch per batch")
    ap.add_argument("--regimes", default="",
                    help="extra (batches per launch)x(launches in flight) combinations measured after the timed region and "
                         "reported under roofline.regimes, e.g. 4x1,8x2,8x2u (u = not gated); the word none = not even the "
                         "two default ones (1x1 and 1x3u)")
    ap.add_argument("--extras", default=os.environ.get("VSS_BENCH_EXTRAS", "auto"),
                    help="other configurations measured after the headline and attached to the same JSON line as objects, each "
                         "in a process of its own: comma list of c2,c4,c5,a13; auto = all of them on the full single-GPU c3 run, "
                         "none = just the headline")
    ap.add_argument("--extras-budget-s", type=float, default=1320.0, help="no extra is started once the run is this old")
    ap.add_argument("--quality-rows", type=int, default=200_000, help="--config quality: rows of the prefix both builds index")
    ap.add_argument("--quality-options", default="", help="--config quality: M/ef_construction pairs, e.g. 16/128,32/384 (default: both)")
    ap.add_argument("--quality-efs", default="", help="--config quality: the ef_search grid, comma separated")
    ap.add_argument("--config", default="c3", choices=["c3", "c2", "c4", "c5", "a13", "quality"],
                    help="c3 = BASELINE configs[2] (default; configs[3] when --gpus > 1), c2 = configs[1] single-query scan, "
                         "c4 = configs[3] at full workload as --shards row-range shards co-resident on ONE GPU (no xGMI), "
                         "c5 = one shard (12.5M rows) of configs[4] with its delete / insert / compact steps")
    ap.add_argument("--shards", type=int, default=8, help="--config c4: row-range shards placed on the one GPU")
    ap.add_argument("--reorder-after-build", action="store_true",
                    help="finish the bulk build with the reference's (level, cluster) compaction order (vss_set_build_reorder); its "
                         "time is part of build_s")
    ap.add_argument("--host-api-seconds", type=float, default=2.0,
                    help="seconds of the concurrent host-pointer vss_search_batch leg (PCIe-inclusive, reported beside value)")
    ap.add_argument("--mode", default="sharded", choices=["sharded", "replicated"],
                    help="N>1: row-range shards + RCCL all-gather merge (configs[3], strong scaling) or one full "
                         "index per GPU with its own query batches (throughput mode, weak scaling, no collective)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-sample-rows", type=int, default=200_000)
    ap.add_argument("--cpu-mt-build-rows", type=int, default=300_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small-launches", action="store_true", help="skip the one-query / join-chunk latency leg")
    ap.add_argument("--repeats", type=int, default=5,
                    help="after the timed region, run the same number of steps this many more times on other batches and report "
                         "[min, median, max] queries/s (`value` stays the first region)")
    ap.add_argument("--wide-ef-sweep", action="store_true",
                    help="let the ef_search sweep go on beyond 512 (640 ... 1536: candidate lists in HBM, MemList) — the reference-"
                         "default index needs that to reach the target recall, if it reaches it at all")
    ap.add_argument("--build-only", action="store_true", help="stop after the bulk build and report rows/s with its roofline")
    ap.add_argument("--sidecar", default=None,
                    help="file the complete result object is written to (default gpurun_out/bench_full_<config>.json); stdout "
                         "carries small detail lines and, LAST, the compact headline line (<= %d characters)" % LINE_LIMIT)
    ap.add_argument("--cpu-prefix-only", action="store_true", help="CPU baseline on a prefix index even if RAM allows the full one")
    args = ap.parse_args()
    t_run0 = time.perf_counter()
    if args.config == "a13":
        return main_a13(args)
    if args.config == "quality":
        return main_quality(args)
    if args.steps is None:
        args.steps = 4000 if args.config == "c2" else 128
    if args.warmup is None:
        args.warmup = 16 if args.config == "c2" else 24
    if args.config == "c2":
        return main_c2(args)
    if args.config == "c5":
        return main_c5(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:  # noqa: E129
        sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d ..." %
                 (args.gpus, args.gpus))
    co_resident = args.config == "c4"  # configs[3] on ONE GPU: every shard lives on this device, no collective
    if co_resident:
        assert world == 1 and args.gpus == 1, "--config c4 places all shards on one GPU (use --gpus N for one shard per GPU)"
    n_local = max(1, args.shards) if co_resident else 1  # shards held by this rank
    force = os.environ.get("VSS_BENCH_FORCE_COLLECTIVE") == "1"  # dev: run the all-gather + merge path with 1 rank
    sharded = ((world > 1 or force) and args.mode == "sharded") or co_resident
    replicated = (world > 1 or force) and not sharded  # (force: the one-rank RCCL dry run of either mode)
    n_shards = world * n_local if sharded else 1
    metric = args.metric or ("l2sq" if sharded else "cosine")
    # dev/test only: all ranks on GPU 0 with the gloo backend (RCCL refuses two ranks on one device) — lets a 1-GPU box
    # run the real multi-process sharded path end to end (tests/test_gpu_parity.py::test_two_rank_sharded_bench)
    same_device = os.environ.get("VSS_BENCH_SAME_DEVICE") == "1"
    if same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    backend = None
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        backend = "gloo" if same_device else "nccl"
        if same_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    # who runs where: every rank reports its device ordinal and PCI address; one rank per GPU means pairwise distinct
    props = torch.cuda.get_device_properties(device)
    me = {"rank": rank, "device": torch.cuda.current_device(), "name": props.name,
          "pci": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                     getattr(props, "pci_device_id", 0)),
          "uuid": str(getattr(props, "uuid", ""))}
    rank_devices = [me]
    if world > 1 or force:
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, me)
        if not same_device:
            assert len({r["pci"] for r in rank_devices}) == world and dist.get_world_size() == world, \
                "ranks do not sit on %d distinct devices: %s" % (world, rank_devices)

    pkg = load_package()
    M, M0, efc = args.M, (args.M0 or 2 * args.M), args.ef_construction
    dim, k, B = args.dim, args.k, args.batch
    n_total = args.rows
    import importlib.util
    spec = importlib.util.spec_from_file_location("vss_sharded", os.path.join(ROOT, "duckdb-vss_amd", "sharded.py"))
    shardlib = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shardlib)
    gen = Mixture(n_total, dim, metric != "l2sq", device)

    # ---------------------------------------------------------------- build (timed separately: rows/s)
    # local shard s of this rank is global shard rank * n_local + s and owns the row range shard_range(...) gives it
    ranges = [shardlib.shard_range(rank * n_local + s, n_shards, n_total) if sharded else (0, n_total) for s in range(n_local)]
    shards, streams = [], []
    for lo, hi in ranges:
        ix = pkg.GpuIndex(dim, metric, M, M0, efc, 64, device=local_rank)
        st = torch.cuda.Stream(device=device)
        ix.set_stream(st.cuda_stream)
        if args.reorder_after_build:
            ix.set_build_reorder(True)
        ix.reserve(hi - lo)
        shards.append(ix)
        streams.append(st)
    index, stream = shards[0], streams[0]
    n_local_rows = sum(hi - lo for lo, hi in ranges)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for (lo, hi), ix in zip(ranges, shards):
        pos = lo
        while pos < hi:  # rows are generated chunk by chunk straight into the index (inputs resident in HBM)
            ci = pos // CHUNK
            x = gen.rows(DATA_SEED, ci, CHUNK)
            a, b = pos - ci * CHUNK, min(hi, (ci + 1) * CHUNK) - ci * CHUNK
            x = x[a:b].contiguous()
            ids = torch.arange(pos, pos + (b - a), dtype=torch.int64, device=device)
            torch.cuda.synchronize()
            ix.stage_device(ids.data_ptr(), x.data_ptr(), b - a)
            pos += b - a
            del x, ids
    torch.cuda.synchronize()
    t_stage = time.perf_counter() - t0
    t0 = time.perf_counter()
    if n_local == 1:
        index.build_finalize()
    else:  # co-resident shards link at the same time, one host thread and one stream each (the devices of an 8-GPU node would
        import threading  # do the same side by side): the small first batches and the tail of every batch overlap
        errors = []

        def link(ix):
            try:
                ix.build_finalize()
            except Exception as e:  # noqa: BLE001
                errors.append(e)
        ths = [threading.Thread(target=link, args=(ix,)) for ix in shards]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        if errors:
            raise errors[0]
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    build_timing, build_work = {}, {}
    for ix in shards:
        for key, v in ix.timing(reset=True).items():
            build_timing[key] = build_timing.get(key, 0) + v
        for key, v in ix.build_work().items():
            build_work[key] = build_work.get(key, 0) + v
    if world > 1 or force:
        tb = torch.tensor([t_build], device=device)
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        t_build = float(tb.item())

    def build_summary():
        """The `index build rows/sec` half of the metric with what explains it: rows/s depends on the index options through the
        distances a row's insertion computes (ef_construction 384 since round 4: 10.5k per row against 7.1k at 256), so the
        line carries distances and algorithmic bytes per row and phase A's fraction of the HBM peak next to the rate."""
        nbytes = build_work["insert_distances"] * (4 * dim + 4) + build_work["insert_expansions"] * (4 + 4 * M0)
        link_bytes = build_work["link_distances"] * (4 * dim + 4)
        return {"rows_per_s": n_total * (world if replicated else 1) / t_build, "M": M, "ef_construction": efc,
                "distances_per_row": build_work["insert_distances"] / max(1, n_local_rows),
                "link_repair_distances_per_row": build_work["link_distances"] / max(1, n_local_rows),
                "algorithmic_MB_per_row": (nbytes + link_bytes) / max(1, n_local_rows) / 1e6,
                "phase_a_frac_of_hbm": nbytes / max(1e-9, build_timing["build_phase_a_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS,
                "whole_build_frac_of_hbm": (nbytes + link_bytes) / max(1e-9, t_build) / 1e9 / HBM_PEAK_GBS}

    if args.build_only:  # the build half of the metric at other index options (the extra `build_efc256`): no search
        if rank == 0:
            finish({"metric": "index build rows/sec, %d×%d FLOAT %s (M %d, ef_construction %d)" % (n_total, dim, metric, M, efc),
                    "config_id": "build", "value": n_total / t_build, "unit": "rows/s", "n_gpus": world, "steps": 1, "warmup": 0,
                    "ms_per_step": t_build * 1e3, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
                    "build_rows_per_s": n_total / t_build, "build": build_summary(), "build_s": t_build, "stage_s": t_stage,
                    "config": {"workload": "bulk build only", "rows": n_total, "dim": dim, "index_metric": metric, "M": M, "M0": M0,
                               "ef_construction": efc, "shards": n_shards},
                    "roofline": {"bound": "hbm", "kernel": "k_build_phase_a", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                 "achieved": build_summary()["phase_a_frac_of_hbm"] * HBM_PEAK_GBS,
                                 "frac": build_summary()["phase_a_frac_of_hbm"], "traffic": None},
                    "cpu_baseline": None}, args.sidecar)
        return

    # ---------------------------------------------------------------- queries + ground truth
    nqb = args.query_batches
    # sharded: every rank sees the same batches; replicated: every rank has its own
    Q = [gen.rows(QUERY_SEED, i + (nqb * rank if replicated else 0), B) for i in range(nqb)]
    torch.cuda.synchronize()  # (generated on torch's stream, searched on the indexes' own streams)
    lib = pkg.load_library()
    comm_stream = torch.cuda.Stream(device=device) if sharded else None

    def packed_merge(packed, n_sh, nq, kk, od, oi):
        rc = lib.vss_merge_topk_packed_device(packed.data_ptr(), n_sh, nq, kk, od.data_ptr(), oi.data_ptr(), None,
                                              comm_stream.cuda_stream)
        assert rc == 0

    def new_exchange(n_batches):
        """Output buffers of one launch: every local shard's block of the packed exchange + per-shard result counts."""
        px = shardlib.PackedExchange(n_batches, B, k, device, packed_merge, n_local=n_local, always_collective=force)
        px.counts = torch.empty((n_local, n_batches, B), dtype=torch.int32, device=device)
        px.evt = None
        return px

    px1 = new_exchange(1)

    def probe(q, ef, exact=False):
        """One step of the hot path, blocking: batched top-k on every local shard (+ exchange and merge when sharded)."""
        for s, ix in enumerate(shards):
            ix.search_batch_device(q.data_ptr(), B, k, ef, px1.ids(0, s).data_ptr(), px1.dists(0, s).data_ptr(),
                                   px1.counts[s, 0].data_ptr(), exact=exact)
        if not sharded:
            return px1.ids(0), px1.dists(0)
        with torch.cuda.stream(comm_stream):
            md, mi = px1.exchange()
        comm_stream.synchronize()
        return mi[0], md[0]

    t0 = time.perf_counter()
    truth = []
    for q in Q[:2]:
        tk, _ = probe(q, 0, exact=True)
        truth.append(tk.clone())
    torch.cuda.synchronize()
    t_exact = (time.perf_counter() - t0) / 2
    # the exact path by itself (BASELINE configs[2]'s "MFMA distance tile"; SURVEY §8d: flops = 2 B N dim, f32 matrix peak 157.3
    # TFLOP/s): one more batch now that the row norms and every scratch buffer exist — scores, select and re-rank over wall clock
    t0 = time.perf_counter()
    probe(Q[0], 0, exact=True)
    torch.cuda.synchronize()
    t_exact_warm = time.perf_counter() - t0
    exact_info = {"batch_s": t_exact_warm, "first_two_batches_s": t_exact,
                  "tflops_over_wall": 2.0 * B * n_local_rows * dim / t_exact_warm / 1e12,
                  "frac_of_f32_mfma_peak": 2.0 * B * n_local_rows * dim / t_exact_warm / 1e12 / 157.3,
                  "kernel": "k_exact_scores_v4 (persistent 128x128 MFMA tile, LDS-DMA operands) + k_exact_select + k_exact_rerank",
                  "kernel_only_evidence": "profiles/r05h_exact_tile_kernel_only_rocprofv3.txt"}

    # ---------------------------------------------------------------- ef_search: smallest that reaches the target recall
    # (a shard returns its own top-k, so the merged result of G shards reaches the target at a smaller per-shard ef:
    # the sweep starts low and every shard count finds its own operating point — SURVEY §8e "tune, don't assume")
    sweep = [args.ef] if args.ef else (EF_SWEEP + EF_SWEEP_WIDE if args.wide_ef_sweep else EF_SWEEP)

    def recalls_at(e):  # per-query recall@k of the selection batches at ef_search = e
        out = []
        for i in range(len(truth)):
            out += recall_per_query(probe(Q[i], e)[0], truth[i])
        return out

    # selection on batches 0-1 (two standard errors above the target), the REPORTED recall on held-out batches below
    ef, sel_recall, sel_se, sweep_log = select_ef(recalls_at, sweep, args.target_recall)
    if world > 1 or force:  # every rank must use the same ef (comparable work)
        t = torch.tensor([ef], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ef = int(t.item())
    held = []
    for i in range(max(0, args.heldout_batches)):  # batches no selection step has seen (replicated ranks: their own)
        qh_ = gen.rows(QUERY_SEED, 5000 + i + (100 * rank if replicated else 0), B)
        torch.cuda.synchronize()  # (generated on torch's stream, searched on the index's own)
        tk, _ = probe(qh_, 0, exact=True)
        tk = tk.clone()
        held += recall_per_query(probe(qh_, ef)[0], tk)
        del qh_, tk
    recall, recall_se = mean_and_se(held) if held else (sel_recall, sel_se)
    recall_info = {"reported": "held-out" if held else "selection batches (no held-out batches requested)",
                   "heldout": {"batches": max(0, args.heldout_batches), "queries": len(held), "mean": round(recall, 5),
                               "se": round(recall_se, 5)} if held else None,
                   "selection": {"batches": len(truth), "queries": len(truth) * B, "mean": round(sel_recall, 5),
                                 "se": round(sel_se, 5), "rule": "smallest ef of the sweep with mean - 2 se >= target"}}

    # ---------------------------------------------------------------- timed region
    depth = max(1, min(4, args.pipeline))
    G = max(1, min(32, args.coalesce))
    while nqb < depth * G:  # every batch of the launches in flight is a different one (no cache help from repeats)
        Q.append(gen.rows(QUERY_SEED, nqb + (1000 * rank if replicated else 0), B))
        nqb += 1
    torch.cuda.synchronize()
    exchanges = {}  # batches per launch -> one PackedExchange per launch in flight (its own gather / merge buffers)
    q_first = [0]  # the probe stream of run_steps starts at this batch of Q (the repeats of the timed region move it on)

    def run_steps(n_steps, depth, G):
        """n_steps probe batches, G of them per launch of the search engine (one launch per local shard) and `depth`
        launches in flight on the shards' search contexts; returns kernel ms (sum over kernel launches), work counters
        and the number of kernel launches.  Sharded: when a launch has completed, ONE exchange (all-gather of the packed
        per-shard blocks + k-way merge of all its queries) runs on a side stream while the next launches search."""
        pxs = exchanges.setdefault(G, [])
        while len(pxs) < depth:
            pxs.append(new_exchange(G))

        def begin(c, s, b0, b1, px):
            ix = shards[s]
            if G == 1:
                ix.search_begin(c, Q[(q_first[0] + b0) % nqb].data_ptr(), B, k, ef, px.ids(0, s).data_ptr(),
                                px.dists(0, s).data_ptr(), px.counts[s, 0].data_ptr())
            else:
                n = b1 - b0
                ix.search_multi_begin(c, [Q[(q_first[0] + b0 + i) % nqb].data_ptr() for i in range(n)], B, k, ef,
                                      [px.ids(i, s).data_ptr() for i in range(n)], [px.dists(i, s).data_ptr() for i in range(n)],
                                      [px.counts[s, i].data_ptr() for i in range(n)])

        def end(c, s):
            ix = shards[s]
            ix.search_end(c)
            st = ix.last_search_stats()
            return ix.timing()["search_kernel_ms"], int(st[0]), int(st[1])

        def exchange(px):  # on the side stream: the next launches search meanwhile
            with torch.cuda.stream(comm_stream):
                px.exchange()
                px.evt = torch.cuda.Event()
                px.evt.record(comm_stream)

        def settle(px):
            if px.evt is not None:
                px.evt.synchronize()  # the context's previous results have been exchanged
                px.evt = None

        out = shardlib.run_pipelined(n_steps, depth, G, n_local, pxs, begin, end, exchange if sharded else None, settle)
        if sharded:
            comm_stream.synchronize()
        return out

    def collectives_so_far():
        return sum(px.collectives for pxs in exchanges.values() for px in pxs)

    run_steps(args.warmup, depth, G)
    if world > 1 or force:
        dist.barrier()
    torch.cuda.synchronize()
    coll0 = collectives_so_far()
    t0 = time.perf_counter()
    kernel_ms, dists, expans, n_launches = run_steps(args.steps, depth, G)
    torch.cuda.synchronize()
    if world > 1 or force:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    collectives_timed = collectives_so_far() - coll0  # all-gathers actually issued by the timed launches (counted, not assumed)

    # ---- the timed region once more, REPEATS times, each on other batches of the probe stream: the spread of `value` (VERDICT
    # r05: three runs of one tree gave 940.6k / 966.0k / 971.1k and the line carried no dispersion).  `value` stays the FIRST
    # region's — the one the contract describes; the repeats are bracketed the same way (barrier + synchronize on both sides).
    repeat = None
    if args.repeats > 0:
        want = min(args.steps * (args.repeats + 1), 192)  # distinct batches in HBM (3 MiB each), the stream wraps beyond
        while nqb < want:
            Q.append(gen.rows(QUERY_SEED, nqb + (1000 * rank if replicated else 0), B))
            nqb += 1
        torch.cuda.synchronize()
        rates, kernel_fracs = [], []
        for r in range(args.repeats):
            q_first[0] = (args.steps * (r + 1)) % nqb
            if world > 1 or force:
                dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            r_ms, r_d, r_e, r_n = run_steps(args.steps, depth, G)
            torch.cuda.synchronize()
            if world > 1 or force:
                dist.barrier()
            dt = time.perf_counter() - t1
            if world > 1 or force:
                tt = torch.tensor([dt], device=device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            rates.append(args.steps * B * (world if replicated else 1) / dt)
            kernel_fracs.append((r_d * (4 * dim + 4) + r_e * (4 + 4 * M0)) / max(1e-9, r_ms / 1e3) / 1e9 / HBM_PEAK_GBS)
        q_first[0] = 0
        first = args.steps * B * (world if replicated else 1) / elapsed
        srt = sorted(rates)
        repeat = {"queries_per_s": [srt[0], srt[len(srt) // 2], srt[-1]], "runs": rates, "first_region": first,
                  "median_over_value": srt[len(srt) // 2] / first, "frac_per_launch": [min(kernel_fracs), max(kernel_fracs)],
                  "what": "the timed region (%d steps) repeated %d times on other batches of the probe stream; [min, median, max]; "
                          "`value` is the first region" % (args.steps, args.repeats)}

    # The wide sweep went beyond the register lists (ef > 512: candidate lists in HBM) because this index does not reach the target:
    # the SAME timed region once more at ef 512 — the largest limit of the 8-register list's kernels, the operating point the
    # reference-default extra reported until round 5 — so that the line carries both
    at_ef_512 = None
    if args.wide_ef_sweep and ef > 512 and world == 1 and not args.ef:
        ef_chosen, ef = ef, 512  # (run_steps' closures read `ef` when they launch)
        run_steps(args.warmup, depth, G)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        k_ms, k_d, k_e, k_n = run_steps(args.steps, depth, G)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        ef = ef_chosen
        k_bytes = (k_d * (4 * dim + 4) + k_e * (4 + 4 * M0)) / max(1, k_n)
        at_ef_512 = {"ef_search": 512, "queries_per_s": args.steps * B / dt, "avg_kernel_ms": k_ms / max(1, k_n),
                     "frac": k_bytes / max(1e-9, k_ms / max(1, k_n) / 1e3) / 1e9 / HBM_PEAK_GBS, "launches": k_n,
                     "recall_at_10_selection_batches": next((e_["recall"] for e_ in sweep_log if e_["ef"] == 512), None),
                     "distances_per_query": k_d / args.steps / B, "expansions_per_query": k_e / args.steps / B,
                     "kernel": "k_search<.., 8, 768>: 12-wave workgroups, pipelined level search, blocked 8-register list, compact visited sets"}

    def regime(g, p, n_steps, gated=True):
        """The same probe stream under another launch regime (outside the timed region, for context)."""
        for ix in shards:
            ix.set_search_gating(gated)
        run_steps(max(g * p, 3), p, g)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        o_ms, o_d, o_e, o_n = run_steps(n_steps, p, g)
        torch.cuda.synchronize()
        o_wall = (time.perf_counter() - t1) / n_steps
        o_bytes = (o_d * (4 * dim + 4) + o_e * (4 + 4 * M0)) / n_steps  # per batch
        per_launch_s = o_ms / 1e3 / o_n
        for ix in shards:
            ix.set_search_gating(True)
        return {"batches_per_launch": g, "launches_in_flight": p, "gated": gated, "ms_per_step": o_wall * 1e3,
                "queries_per_s": B / o_wall,
                "avg_kernel_ms": per_launch_s * 1e3,
                "gbs_per_launch": o_bytes * n_steps / o_n / per_launch_s / 1e9 if per_launch_s else 0.0,
                "frac_per_launch": o_bytes * n_steps / o_n / per_launch_s / 1e9 / HBM_PEAK_GBS if per_launch_s else 0.0,
                "gbs_over_wall": o_bytes / o_wall / 1e9, "frac_over_wall": o_bytes / o_wall / 1e9 / HBM_PEAK_GBS}

    # outside the timed region, for context: one launch per batch — one probe at a time, and three in flight on separate
    # search contexts (round 1's regime) — and whatever --regimes asks for; then the host-pointer API under concurrent callers
    regimes = []
    if world == 1:
        wanted = [] if (args.regimes == "none" or co_resident) else [(1, 1, True), (1, 3, False)]  # round 1's two figures
        for item in [x for x in args.regimes.split(",") if x and x != "none"]:  # e.g. 8x2 (gated) or 8x2u (issued immediately)
            item = item.lower()
            g, p = (int(v) for v in item.rstrip("u").split("x"))
            wanted.append((max(1, min(32, g)), max(1, min(4, p)), not item.endswith("u")))
        for g, p, gated in wanted:
            if (g, p, gated) != (G, depth, True):
                regimes.append(regime(g, p, 24 if g == 1 else 6 * g, gated))
    host_api = None
    if world == 1 and n_shards == 1 and args.host_api_seconds > 0:
        # HNSW_INDEX_JOIN as DuckDB would drive it: host buffers in, host buffers out (3 MiB H2D + 120 KiB D2H per
        # 1024-query batch), several operator threads probing the same index at once (the engine leases each a context)
        import threading
        n_threads, counts = 4, []
        q_host = [q.cpu().numpy() for q in Q[:4]]

        def session(t):
            done, t_end = 0, time.perf_counter() + args.host_api_seconds
            while time.perf_counter() < t_end:
                index.search_batch(q_host[(t + done) % len(q_host)], k, ef)
                done += 1
            counts.append(done)

        index.search_batch(q_host[0], k, ef)
        threads = [threading.Thread(target=session, args=(t,)) for t in range(n_threads)]
        t1 = time.perf_counter()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        dt = time.perf_counter() - t1
        t1 = time.perf_counter()
        n1 = 0
        while time.perf_counter() - t1 < args.host_api_seconds / 2:
            index.search_batch(q_host[n1 % len(q_host)], k, ef)
            n1 += 1
        dt1 = time.perf_counter() - t1
        host_api = {"threads": n_threads, "queries_per_s": sum(counts) * B / dt, "one_thread_queries_per_s": n1 * B / dt1,
                    "what": "vss_search_batch on host pointers (PCIe-inclusive: queries H2D, ids + distances + counts D2H)"}
    small = None
    if world == 1 and n_shards == 1 and not args.no_small_launches:
        small = small_launches(index, gen, k, ef, max(1, 2048 // k))
    if world > 1 or force:
        te = torch.tensor([elapsed, recall], device=device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te[0].item())
        tr = torch.tensor([recall], device=device)
        dist.all_reduce(tr, op=dist.ReduceOp.MIN)
        recall = float(tr.item())

    # ---------------------------------------------------------------- roofline of the dominant kernel (k_search)
    # algorithmic bytes per query: n_dist * (4*dim + 4) + n_expand * (4 + 4*M0)
    steps = max(1, args.steps)
    n_launches = max(1, n_launches)
    bytes_per_launch = (dists * (4 * dim + 4) + expans * (4 + 4 * M0)) / n_launches
    avg_kernel_s = kernel_ms / 1e3 / n_launches
    achieved = bytes_per_launch / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0

    # HBM traffic of the kernel comes from separate rocprofv3 --pmc passes of this same command (committed under
    # profiles/); it is attached only when that pass was taken on exactly this configuration
    traffic, traffic_src = None, None
    try:
        import glob
        per_launch = steps * n_local / n_launches  # batches per launch in the timed region (the last launch may be shorter)
        # (only a pass counted on THIS launch shape is attached — cache reuse grows with the batches a launch
        #  carries, so a figure scaled from another shape is not this run's; files of superseded trees ("before") are skipped;
        #  among several passes of the shape the newest round's wins: the paths sort by round)
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_k_search*.json"))):
            if "before" in os.path.basename(path):
                continue
            pm = json.load(open(path))
            c = pm["config"]
            if (c["rows"], c["dim"], c["index_metric"], c["M"], c["M0"], c["ef_construction"], c["ef_search"],
                    c["batch_queries"], c["k"]) == (n_total, dim, metric, M, M0, efc, ef, B, k) and world == 1 and \
                    c.get("shards", 1) == n_shards and pm.get("batches_per_launch", 1) == per_launch:
                traffic = pm["hbm_bytes_per_launch"]
                traffic_src = os.path.relpath(path, ROOT) + " (committed rocprofv3 --pmc passes of this command, not this run's counters)"
    except Exception:
        pass

    result = None
    if rank == 0:
        full = (n_total == 10_000_000 and dim == 768 and B == 1024 and k == 10)
        where = ("%d row-range shards co-resident on ONE MI355X (no xGMI, no collective: per-shard results merged by the packed "
                 "k-way merge kernel)" % n_shards if co_resident else
                 "row-range sharded over %d MI355X + one RCCL all-gather per launch + merge" % world if sharded else
                 "replicated on %d MI355X, one 1024-query batch stream per GPU" % world if replicated else "single MI355X")
        workload = ("configs[%d]: 10M rows FLOAT[768] %s top-10, batched 1024 queries, %s" %
                    (3 if sharded else 2, metric, where)) if full else \
            "DEVELOPMENT RUN (not the benchmark): %d rows FLOAT[%d] %s top-%d, batch %d, %s" % (n_total, dim, metric, k, B, where)
        result = {
            "metric": "queries/sec at recall@10, 10M×768 FLOAT top-10; index build rows/sec",  # BASELINE.json, verbatim
            "config_id": "c4" if co_resident else "c3",
            "value": args.steps * B * (world if replicated else 1) / elapsed, "unit": "queries/s", "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if (sharded and not co_resident) else "weak",
            "multi_gpu_mode": args.mode if (world > 1 or force) else None, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "recall_at_10": round(recall, 4), "recall_at_10_se": round(recall_se, 5), "recall": recall_info,
            "ef_search": ef, "ef_sweep": sweep_log,
            "build_rows_per_s": n_total * (world if replicated else 1) / t_build, "build_s": t_build, "stage_s": t_stage,
            "build": build_summary(),
            "build_kernel_ms": {"phase_a": build_timing["build_phase_a_ms"], "phase_b": build_timing["build_phase_b_ms"],
                                "batches": build_timing["build_batches"], "retries": build_timing["build_retries"]},
            "build_roofline": {
                "bound": "hbm", "kernel": "k_build_phase_a", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "algorithmic_bytes": build_work["insert_distances"] * (4 * dim + 4) + build_work["insert_expansions"] * (4 + 4 * M0),
                "achieved": (build_work["insert_distances"] * (4 * dim + 4) + build_work["insert_expansions"] * (4 + 4 * M0)) /
                            max(1e-9, build_timing["build_phase_a_ms"] / 1e3) / 1e9,
                "distances_per_row": build_work["insert_distances"] / max(1, n_local_rows),
                "link_repair_distances_per_row": build_work["link_distances"] / max(1, n_local_rows)},
            "at_ef_512": at_ef_512,
            "repeat": ({k_: repeat[k_] for k_ in ("queries_per_s", "median_over_value")} if repeat else None),
            "repeat_detail": repeat,
            "plateau": (None if recall >= args.target_recall or not args.wide_ef_sweep else
                        {"engine": round(recall, 4), "at_ef_search": ef, "reference": None,
                         "note": "no ef_search of the sweep reaches the target on this 10M-row index; a reference-built 10M-row graph "
                                 "is hours of host build — at 200k rows (`quality` extra) and at 1M rows "
                                 "(profiles/r06b_quality_1m768_default_options.json) of this data the reference-built and the "
                                 "engine-built graph of these options answer within 0.006 of each other at every ef_search"}),
            "exact_batch_s": t_exact, "exact": exact_info,
            "host_api": host_api, "host_api_queries_per_s": host_api["queries_per_s"] if host_api else None,
            "small_launches": small,
            "rccl_ranks": world if backend == "nccl" else 0, "collective_backend": backend,
            "collectives_per_launch": collectives_timed / max(1, n_launches // n_local), "collectives_timed": collectives_timed,
            "rank_devices": rank_devices, "rank_pci": [r["pci"] for r in rank_devices],
            "recall_measured_on": "%d held-out queries vs the exact MFMA path" % recall_info["heldout"]["queries"]
                                  if recall_info.get("heldout") else "the selection batches (no held-out batches requested)",
            "config": {"workload": workload, "rows": n_total, "dim": dim, "index_metric": metric, "k": k,
                       "batch_queries": B, "M": M, "M0": M0, "ef_construction": efc, "ef_search": ef,
                       "batches_per_launch": G, "batches_per_launch_timed": steps * n_local / n_launches,
                       "launches_in_flight": depth, "launches_gated": True, "shards": n_shards,
                       "reordered_after_build": bool(args.reorder_after_build),
                       "shards_per_gpu": n_local,
                       "parallelism": ("shard%d-on-1-gpu" % n_shards if co_resident else "shard%d" % world if sharded else
                                       "replica%d" % world if replicated else "single")},
            "roofline": {"bound": "hbm", "kernel": "k_search", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_over_algorithmic": traffic / bytes_per_launch if traffic and bytes_per_launch else None,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_kernel_ms": avg_kernel_s * 1e3,
                         "launches": n_launches,
                         "effective_gbs_over_wall": bytes_per_launch * n_launches / elapsed / 1e9,
                         "frac_over_wall": bytes_per_launch * n_launches / elapsed / 1e9 / HBM_PEAK_GBS,
                         "regimes": regimes,
                         "distances_per_query": dists / steps / B, "expansions_per_query": expans / steps / B,
                         "visited_set": visited_set_form(max(k, ef), n_total // n_shards)},
        }
    # the CPU baseline runs on rank 0 at N=1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        def gpu_answer(qs):  # the engine's answers to the reference's queries: same graph(s), same ef, through the device path
            keys, ds = [], []
            for b0 in range(0, len(qs), B):
                qb = torch.from_numpy(np.ascontiguousarray(qs[b0:b0 + B])).to(device)
                if len(qb) < B:
                    qb = torch.cat([qb, qb[:1].expand(B - len(qb), dim)]).contiguous()
                mi, md = probe(qb, ef)
                torch.cuda.synchronize()
                keys.append(mi.cpu().numpy().copy())
                ds.append(md.cpu().numpy().copy())
            return np.concatenate(keys)[:len(qs)], np.concatenate(ds)[:len(qs)]

        result["cpu_baseline"] = cpu_baseline(pkg, args, gen, dim, metric, k, ef, device, M, M0, efc, shards=shards,
                                              gpu_answer=gpu_answer)
        if small and result["cpu_baseline"]["value"]:  # the reference thread beside the two latency shapes (same graph, same ef)
            ref_us = 1e6 / result["cpu_baseline"]["value"]
            small["single_query"]["reference_thread_us_per_call"] = ref_us
            small["join_chunk"]["reference_thread_us_per_call"] = ref_us * small["join_chunk"]["queries"]
    # the other BASELINE configurations, each in a process of its own, once this one has let go of its index (HBM and host RAM)
    extras = []
    if rank == 0 and world == 1 and not force and not co_resident:
        full_run = (n_total == 10_000_000 and dim == 768 and B == 1024 and k == 10)
        extras = ([] if args.extras == "none" else DEFAULT_EXTRAS if args.extras == "auto" and full_run else
                  [] if args.extras == "auto" else [x for x in args.extras.split(",") if x in EXTRA_CONFIGS])
    extra_objects = []
    if extras:
        # the headline first, before anything else can go wrong: should an extra hang past the caller's patience, the last
        # complete JSON line on stdout is still the headline (it is printed once more, LAST, when the extras are done)
        print(json.dumps(compact_line(result)))
        sys.stdout.flush()
        for ix in shards:
            ix.close()
        del shards, index, Q, truth, px1, exchanges
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        extra_objects = list(run_extras(extras, args.extras_budget_s, t_run0).items())
        result["extras"] = {"configs": extras, "run_wall_s": round(time.perf_counter() - t_run0, 1),
                            "what": "the other BASELINE configurations (and row a13), one `python bench.py --config ...` process each, "
                                    "after the headline index was released; one {\"extra\": ...} line each on stdout, complete objects "
                                    "in the sidecar file"}
        for name, obj in extra_objects:  # complete objects: sidecar only
            result[name] = obj
    failed = False
    if rank == 0:
        emit(result, args.sidecar, extra_objects)
        a = (result.get("cpu_baseline") or {}).get("agreement")
        if a is not None and not agreement_ok(a):
            sys.stderr.write("bench.py: reference agreement below the bar: %s\n" % json.dumps(a))
            failed = True
        for name in extras:  # an extra whose own agreement gate failed fails the run as well (its line is printed either way)
            if result[name].get("exit_code") == 4:
                sys.stderr.write("bench.py: reference agreement below the bar in %s\n" % name)
                failed = True
    if world > 1 or force:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit(4)


if __name__ == "__main__":
    main()
