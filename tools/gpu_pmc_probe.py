"""Small search workload for rocprofv3 --pmc runs (not a pytest module): build 200k x 768, run one 1024-query batch."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dim, metric, B, k, ef = 768, "cosine", int(sys.argv[2]) if len(sys.argv) > 2 else 1024, 10, 64
pkg = load_package()
dev = torch.device("cuda", 0)
gen = bench.Mixture(rows, dim, True, dev)
idx = pkg.GpuIndex(dim, metric)
idx.reserve(rows)
x = gen.rows(bench.DATA_SEED, 0, rows)
ids = torch.arange(rows, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
idx.stage_device(ids.data_ptr(), x.data_ptr(), rows)
idx.build_finalize()
q = gen.rows(bench.QUERY_SEED, 0, B)
ok = torch.empty((B, k), dtype=torch.int64, device=dev)
od = torch.empty((B, k), dtype=torch.float32, device=dev)
oc = torch.empty(B, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for _ in range(3):
    idx.search_batch_device(q.data_ptr(), B, k, ef, ok.data_ptr(), od.data_ptr(), oc.data_ptr())
print("search kernel ms", idx.timing()["search_kernel_ms"], idx.last_search_stats())
