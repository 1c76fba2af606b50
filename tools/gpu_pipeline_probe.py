"""Throughput vs batches-in-flight on one resident index (not a pytest module).
    python tools/gpu_pipeline_probe.py rows M ef [ef ...]"""
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

rows, M = int(sys.argv[1]), int(sys.argv[2])
efs = [int(x) for x in sys.argv[3:]] or [320]
dim, metric, B, k = 768, "cosine", 1024, 10
pkg = load_package()
dev = torch.device("cuda", 0)
gen = bench.Mixture(rows, dim, True, dev)
idx = pkg.GpuIndex(dim, metric, M, 2 * M, 128, 64)
idx.reserve(rows)
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
print("build %.1fs %s" % (time.time() - t0, idx.timing(reset=True)))
Q = [gen.rows(bench.QUERY_SEED, i, B) for i in range(8)]
slots = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
          torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(4)]
torch.cuda.synchronize()
for ef in efs:
    for depth in (1, 2, 3, 4):
        steps = 24
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            kms = 0.0
            for i in range(steps + depth):
                c = i % depth
                if i >= depth:
                    idx.search_end(c)
                    kms += idx.timing()["search_kernel_ms"]
                if i < steps:
                    a, b, cc = slots[c]
                    idx.search_begin(c, Q[i % 8].data_ptr(), B, k, ef, a.data_ptr(), b.data_ptr(), cc.data_ptr())
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        st = idx.last_search_stats()
        print("ef=%d depth=%d: %.0f qps, %.3f ms/step wall, kernel %.3f ms avg, dists/q %.0f, retried %d" % (
            ef, depth, steps * B / el, el / steps * 1e3, kms / steps, st[0] / B, st[3]))
