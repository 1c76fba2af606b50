"""k_merge_topk at the shapes the sharded probe produces (not a pytest module): n_shards x k cells per query, `nq` queries per
call — one launch of up to 32 batches of 1024 queries is 32 768 queries.  Inputs as the engine writes them: ascending per shard,
random row ids; events on the stream the kernel runs on.  Prints one JSON line per shape (gpurun_out/merge_probe.jsonl)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
lib = pkg.load_library()
dev = torch.device("cuda", 0)
shapes = [(8, 10, 32768), (8, 100, 32768), (8, 100, 1024), (8, 2047, 1024), (3, 1, 32768), (2, 100, 32768), (4, 100, 32768)]
out_path = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(HERE)), "gpurun_out", "merge_probe.jsonl")
os.makedirs(os.path.dirname(out_path), exist_ok=True)
with open(out_path, "a") as f:
    for G, k, nq in shapes:
        g = torch.Generator(device=dev).manual_seed(G * 100000 + k)
        d = torch.sort(torch.rand((G, nq, k), generator=g, device=dev), dim=2).values.contiguous()
        ids = torch.randperm(G * nq * k, generator=g, device=dev).view(G, nq, k).contiguous()
        od = torch.empty((nq, k), dtype=torch.float32, device=dev)
        oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            assert lib.vss_merge_topk_device(d.data_ptr(), ids.data_ptr(), G, nq, k, od.data_ptr(), oi.data_ptr(), None, st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            lib.vss_merge_topk_device(d.data_ptr(), ids.data_ptr(), G, nq, k, od.data_ptr(), oi.data_ptr(), None, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        # spot check against a sort of the union
        fd = d[:, :64].permute(1, 0, 2).reshape(64, G * k)
        fi = ids[:, :64].permute(1, 0, 2).reshape(64, G * k)
        # ascending (distance, row id): sort by id first, then stably by distance
        by_id = torch.sort(fi, dim=1).indices
        order = torch.gather(by_id, 1, torch.sort(torch.gather(fd, 1, by_id), dim=1, stable=True).indices)[:, :k]
        ok = bool(torch.equal(torch.gather(fi, 1, order), oi[:64]))
        line = {"n_shards": G, "k": k, "queries": nq, "ms_per_call": round(ms, 4), "us_per_query": round(ms * 1e3 / nq, 4),
                "cells_per_query": G * k, "bytes_in": G * nq * k * 12, "GBs_in": round(G * nq * k * 12 / ms / 1e6, 1),
                "staged_in_lds": bool(16 + G * k * 12 <= 48 * 1024), "first_64_queries_equal_a_sort_of_the_union": ok}
        print(json.dumps(line))
        f.write(json.dumps(line) + "\n")
