"""Recall vs index size / build batch size on bench-shaped data (not a pytest module).
    python tools/gpu_recall_scan.py rows [max_batch growth_div [M M0 efc]]"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

rows = int(sys.argv[1])
mb, gd = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16384, 32)
M, M0, efc = (int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (16, 32, 128)
dim, metric, B, k = 768, "cosine", 1024, 10
pkg = load_package()
dev = torch.device("cuda", 0)
gen = bench.Mixture(rows, dim, True, dev)
idx = pkg.GpuIndex(dim, metric, M, M0, efc, 64)
idx.reserve(rows)
idx.set_build_params(mb, gd)
pos = 0
while pos < rows:
    n = min(bench.CHUNK, rows - pos)
    x = gen.rows(bench.DATA_SEED, pos // bench.CHUNK, bench.CHUNK)[:n].contiguous()
    ids = torch.arange(pos, pos + n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    idx.stage_device(ids.data_ptr(), x.data_ptr(), n)
    pos += n
t0 = time.time()
idx.build_finalize()
tb = time.time() - t0
q = gen.rows(bench.QUERY_SEED, 0, B)
ok = torch.empty((B, k), dtype=torch.int64, device=dev)
od = torch.empty((B, k), dtype=torch.float32, device=dev)
oc = torch.empty(B, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
idx.search_batch_device(q.data_ptr(), B, k, 0, ok.data_ptr(), od.data_ptr(), oc.data_ptr(), exact=True)
truth = ok.clone()
out = []
for ef in (64, 128, 256, 512):
    idx.search_batch_device(q.data_ptr(), B, k, ef, ok.data_ptr(), od.data_ptr(), oc.data_ptr())
    idx.search_batch_device(q.data_ptr(), B, k, ef, ok.data_ptr(), od.data_ptr(), oc.data_ptr())
    st = idx.last_search_stats()
    out.append("ef%d: r=%.3f %.0fd %.2fms" % (ef, bench.recall_at_k(ok, truth), st[0] / B, idx.timing()["search_kernel_ms"]))
s0 = idx.level_stats(0)
print("rows %d id=%d batch %d/%d M=%d/%d efc=%d: build %.1fs (%.0f rows/s) deg0 %.1f maxlevel %d | %s" % (
    rows, bench.INTRINSIC_DIM, mb, gd, M, M0, efc, tb, rows / tb, s0[1] / s0[0], idx.max_level(), "  ".join(out)))
