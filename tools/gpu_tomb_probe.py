"""Searches over tombstones at an ef beyond 256 (configs[4] shape: ip, dim 1536, top-100): register queue against the
unbounded queue in HBM.   python tools/gpu_tomb_probe.py [rows] [dim] [ef]"""
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

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
ef = int(sys.argv[3]) if len(sys.argv) > 3 else 448
k, B, M, efc = 100, 1024, 32, 128
dev = torch.device("cuda", 0)
pkg = load_package()
gen = bench.Mixture(rows, dim, True, dev)
idx = pkg.GpuIndex(dim, "ip", M, 2 * M, efc, ef)
idx.reserve(rows)
for c in range(0, rows, bench.CHUNK):
    m = min(bench.CHUNK, rows - c)
    x = gen.rows(bench.DATA_SEED, c // bench.CHUNK, m)
    ids = torch.arange(c, c + m, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    idx.stage_device(ids.data_ptr(), x.data_ptr(), m)
    del x, ids
t0 = time.perf_counter()
idx.build_finalize()
print("build %.2f s" % (time.perf_counter() - t0), flush=True)
Q = gen.rows(bench.QUERY_SEED, 0, B)
ok = torch.empty((B, k), dtype=torch.int64, device=dev)
od = torch.empty((B, k), dtype=torch.float32, device=dev)
oc = torch.empty(B, dtype=torch.int32, device=dev)


def run(tag):
    for _ in range(2):
        idx.search_batch_device(Q.data_ptr(), B, k, ef, ok.data_ptr(), od.data_ptr(), oc.data_ptr())
    torch.cuda.synchronize()
    ms = idx.timing()["search_kernel_ms"]
    st = idx.last_search_stats()
    print("%-40s %.2f ms per %d-query batch; %d retried" % (tag, ms, B, int(st[3])), flush=True)
    return ok.cpu().numpy().copy(), od.cpu().numpy().copy()


run("no tombstones")
g = torch.Generator(device="cpu").manual_seed(1234)
dead = torch.randperm(rows, generator=g)[:rows // 100].numpy().astype(np.int64)
idx.remove(dead)
a = run("1 %% tombstones, ef %d (%s)" % (ef, os.environ.get("VSS_SEARCH_REG_QUEUE_MAX", "default queue")))
np.save(os.environ.get("TOMB_OUT", "/tmp/tomb_out.npy"), a[0])
