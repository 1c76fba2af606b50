"""Launch latency of the search engine's shapes over batch sizes (not a pytest module): the workgroup engine (mailboxes), the
one-wave solo shape and teams (walker + helper waves behind workgroup barriers), every launch on queries it has not seen.
    python tools/gpu_team_probe.py [rows] [dim] [metric] [M] [efc] [ef]"""
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

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
metric = sys.argv[3] if len(sys.argv) > 3 else "l2sq"
M = int(sys.argv[4]) if len(sys.argv) > 4 else 16
efc = int(sys.argv[5]) if len(sys.argv) > 5 else 128
ef = int(sys.argv[6]) if len(sys.argv) > 6 else 64
pkg = load_package()
dev = torch.device("cuda", 0)
gen = bench.Mixture(rows, dim, metric != "l2sq", dev)
idx = pkg.GpuIndex(dim, metric, M, 2 * M, efc)
idx.reserve(rows)
t0 = time.perf_counter()
for c in range(0, rows, bench.CHUNK):
    m = min(bench.CHUNK, rows - c)
    x = gen.rows(bench.DATA_SEED, c // bench.CHUNK, m)
    ids = torch.arange(c, c + m, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    idx.stage_device(ids.data_ptr(), x.data_ptr(), m)
idx.build_finalize()
torch.cuda.synchronize()
print("built %d x %d %s in %.1f s" % (rows, dim, metric, time.perf_counter() - t0), flush=True)
k, NQ = 10, 8192
Qall = gen.rows(bench.QUERY_SEED, 0, NQ)
shapes = (("engine", 0, True), ("one wave", 2, False), ("team", 2, True), ("auto", 1, True))
for B in (1, 8, 32, 64, 128, 256):
    reps = max(8, min(64, NQ // (3 * B)))
    line, answers = [], {}
    for si, (name, mode, team) in enumerate(shapes):
        idx.set_search_solo(mode)
        idx.set_search_team(team)
        ok = torch.empty((reps, B, k), dtype=torch.int64, device=dev)
        od = torch.empty((reps, B, k), dtype=torch.float32, device=dev)
        oc = torch.empty((reps, B), dtype=torch.int32, device=dev)
        kms = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):  # every shape gets the same queries, but never one it (or the caches) saw in the previous launch
            q = Qall[((r * len(shapes) + si) * B) % (NQ - B):][:B]
            idx.search_batch_device(q.data_ptr(), B, k, ef, ok[r].data_ptr(), od[r].data_ptr(), oc[r].data_ptr())
            kms += idx.timing()["search_kernel_ms"]
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        line.append("%s %.1f us (kernel %.1f)" % (name, wall * 1e6, kms / reps * 1e3))
    # the same queries through every shape: identical answers
    q = Qall[:B]
    for name, mode, team in shapes:
        idx.set_search_solo(mode)
        idx.set_search_team(team)
        ok1 = torch.empty((B, k), dtype=torch.int64, device=dev)
        od1 = torch.empty((B, k), dtype=torch.float32, device=dev)
        oc1 = torch.empty(B, dtype=torch.int32, device=dev)
        idx.search_batch_device(q.data_ptr(), B, k, ef, ok1.data_ptr(), od1.data_ptr(), oc1.data_ptr())
        torch.cuda.synchronize()
        answers[name] = (ok1.cpu().numpy(), od1.cpu().numpy().view(np.uint32))
    same = all(np.array_equal(answers["engine"][0], a[0]) and np.array_equal(answers["engine"][1], a[1]) for a in answers.values())
    print("B=%4d  " % B + " | ".join(line) + " | identical answers: %s" % same, flush=True)
