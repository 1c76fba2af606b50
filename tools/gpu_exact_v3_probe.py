"""Round 5: the exact path's score tile as persistent workgroups (k_exact_scores_v3, VSS_EXACT_KERNEL=4) against round 3/4's
one-tile workgroups (k_exact_scores_v2, =2): 1024 queries x rows x 768 cosine, seconds per batch and TFLOP/s over wall clock
(scores + select + re-rank), with the select folded into the tile (default) and the plain way, v3 with and without the half-tile
stagger of a compute unit's second workgroup (VSS_EXACT_PROBE=8); ids, distance bits and counts must be identical throughout.
    python tools/gpu_exact_v3_probe.py [rows]"""
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
VARIANTS = [("v2 (round 4), select folded", dict(VSS_EXACT_KERNEL="2", VSS_EXACT_FILTER="1")),
            ("v3 persistent, select folded", dict(VSS_EXACT_KERNEL="4", VSS_EXACT_FILTER="1")),
            ("v4 persistent + LDS-DMA, folded", dict(VSS_EXACT_KERNEL="5", VSS_EXACT_FILTER="1")),
            ("v4, no stagger", dict(VSS_EXACT_KERNEL="5", VSS_EXACT_FILTER="1", VSS_EXACT_PROBE="8")),
            ("v4 persistent + LDS-DMA, plain", dict(VSS_EXACT_KERNEL="5", VSS_EXACT_FILTER="0")),
            ("v2 (round 4), plain select", dict(VSS_EXACT_KERNEL="2", VSS_EXACT_FILTER="0")),
            ("v3 persistent, plain select", dict(VSS_EXACT_KERNEL="4", VSS_EXACT_FILTER="0"))]
if os.environ.get("VSS_PROBE_CHILD") is None:
    outs = []
    for i, (name, env) in enumerate(VARIANTS):
        p = subprocess.run([sys.executable, __file__, str(rows)], env=dict(os.environ, VSS_PROBE_CHILD=str(i), **env),
                           capture_output=True, text=True)
        print("%-34s %s" % (name, p.stdout.strip() or p.stderr[-800:]), flush=True)
        outs.append(np.load("/tmp/exact_v3_probe_%d.npz" % i))
    same = all(np.array_equal(outs[0][k], o[k]) for o in outs[1:] for k in ("keys", "bits", "counts"))
    print("identical answers (ids, distance bits, counts) in all %d variants: %s" % (len(outs), same))
    sys.exit(0 if same else 1)

import torch  # noqa: E402
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

dim, B, k = 768, 1024, 10
pkg = load_package()
dev = torch.device("cuda", 0)
gen = bench.Mixture(rows, dim, True, dev)
idx = pkg.GpuIndex(dim, "cosine", 8, 16, 16)  # a cheap graph: only the exact path is timed
idx.reserve(rows)
for c in range(0, rows, bench.CHUNK):
    m = min(bench.CHUNK, rows - c)
    x = gen.rows(bench.DATA_SEED, c // bench.CHUNK, m)
    ids = torch.arange(c, c + m, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    idx.stage_device(ids.data_ptr(), x.data_ptr(), m)
idx.build_finalize()
q = gen.rows(bench.QUERY_SEED, 0, B)
ok = torch.empty((B, k), dtype=torch.int64, device=dev)
od = torch.empty((B, k), dtype=torch.float32, device=dev)
oc = torch.empty(B, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
best = 1e9
for r in range(5):
    t0 = time.perf_counter()
    idx.search_batch_device(q.data_ptr(), B, k, 0, ok.data_ptr(), od.data_ptr(), oc.data_ptr(), exact=True)
    torch.cuda.synchronize()
    if r:
        best = min(best, time.perf_counter() - t0)
flop = 2.0 * B * rows * dim
print("%d rows: %.4f s per 1024-query batch = %.1f TFLOP/s over wall clock = %.3f of the f32 matrix peak (157.3)" % (
    rows, best, flop / best / 1e12, flop / best / 1e12 / 157.3))
np.savez("/tmp/exact_v3_probe_%s.npz" % os.environ["VSS_PROBE_CHILD"], keys=ok.cpu().numpy(), bits=od.cpu().numpy().view(np.uint32),
         counts=oc.cpu().numpy())
