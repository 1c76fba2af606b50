"""Round 4: fewer bytes per query at the same recall.  For each index option set (M, M0, ef_construction)
build the headline index (rows x 768 cosine), pick the operating point with bench.py's own rule (smallest ef_search whose
selection recall clears 0.95 by two standard errors) and report, on held-out queries: recall, distances and expansions per
query, ALGORITHMIC BYTES PER QUERY (= n_dist * (4 dim + 4) + n_expand * (4 + 4 M0)), plus what the build cost.
    python tools/gpu_option_sweep.py [rows] M:M0:efc [M:M0:efc ...]      -> one JSON line per option set"""
import json
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

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sets = [tuple(int(v) for v in a.split(":")) for a in sys.argv[2:]] or [(32, 64, 256)]
dim, metric, k, B = 768, "cosine", 10, 1024
pkg = load_package()
dev = torch.device("cuda", 0)
gen = bench.Mixture(rows, dim, True, dev)
sel_q = [gen.rows(bench.QUERY_SEED, i, B) for i in range(2)]
held_q = [gen.rows(bench.QUERY_SEED, 5000 + i, B) for i in range(4)]
torch.cuda.synchronize()
outs = (torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
        torch.empty(B, dtype=torch.int32, device=dev))
sweep = [16, 24, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 128, 144, 160, 192, 224, 256, 320, 384, 448, 512]
for M, M0, efc in sets:
    idx = pkg.GpuIndex(dim, metric, M, M0, efc)
    idx.reserve(rows)
    for c in range(0, rows, bench.CHUNK):
        x = gen.rows(bench.DATA_SEED, c // bench.CHUNK, bench.CHUNK)[:min(bench.CHUNK, rows - c)].contiguous()
        ids = torch.arange(c, c + len(x), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        idx.stage_device(ids.data_ptr(), x.data_ptr(), len(x))
        del x, ids
    t0 = time.perf_counter()
    idx.build_finalize()
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    work = idx.build_work()

    def answer(q, ef, exact=False):
        idx.search_batch_device(q.data_ptr(), B, k, ef, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), exact=exact)
        torch.cuda.synchronize()
        return outs[0].clone()

    truth_sel = [answer(q, 0, True) for q in sel_q]
    truth_held = [answer(q, 0, True) for q in held_q]

    def recalls_at(e):
        out = []
        for q, t in zip(sel_q, truth_sel):
            out += bench.recall_per_query(answer(q, e), t)
        return out

    ef, sel_mean, sel_se, log = bench.select_ef(recalls_at, sweep, 0.95)
    held, nd, ne, kms = [], 0, 0, 0.0
    for q, t in zip(held_q, truth_held):
        held += bench.recall_per_query(answer(q, ef), t)
        st = idx.last_search_stats()
        nd, ne, kms = nd + int(st[0]), ne + int(st[1]), kms + idx.timing()["search_kernel_ms"]
    nq = len(held_q) * B
    mean, se = bench.mean_and_se(held)
    bytes_q = nd / nq * (4 * dim + 4) + ne / nq * (4 + 4 * M0)
    print(json.dumps({"rows": rows, "M": M, "M0": M0, "ef_construction": efc, "ef_search": ef,
                      "reached_target": bool(sel_mean - 2 * sel_se >= 0.95),
                      "recall_selection": round(sel_mean, 4), "recall_heldout": round(mean, 4), "recall_heldout_se": round(se, 5),
                      "distances_per_query": nd / nq, "expansions_per_query": ne / nq, "algorithmic_bytes_per_query": bytes_q,
                      "one_batch_launch_ms": kms / len(held_q), "build_s": build_s, "build_rows_per_s": rows / build_s,
                      "build_distances_per_row": (work["insert_distances"] + work["link_distances"]) / rows,
                      "graph_bytes_per_row": 4 * M0 + 4 * M / (M - 1.0), "ef_sweep": log}), flush=True)
    idx.close()
    del idx
    torch.cuda.empty_cache()
