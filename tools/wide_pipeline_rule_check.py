"""CPU: the successor rule of the engine's pipelined level search checked by the oracle's model
(orc_set_pipeline_check) over a random option space — M, M0 <= 64, ef_construction, dimension, metric, tie-free data / an
integer lattice / zero vectors, deletions — far beyond the collected test's four cases.
    python tools/wide_pipeline_rule_check.py [first_seed n_seeds]      -> expansions checked, left to the plain order, WRONG"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import datagen  # noqa: E402
from oracle_lib import CpuIndex, load_oracle  # noqa: E402

first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 200)
orc = load_oracle()
tot = np.zeros(3, dtype=np.int64)
for seed in range(first, first + count):
    rng = np.random.default_rng(70_000 + seed)
    M = int(rng.integers(2, 33))
    M0 = int(rng.integers(M, 65))
    d = int(rng.choice([1, 2, 3, 5, 16, 33, 100]))
    metric = ["l2sq", "cosine", "ip"][int(rng.integers(3))]
    efc = int(rng.integers(1, 121))
    n = int(rng.integers(200, 3000))
    kind = int(rng.integers(3))
    X = datagen.mixture(n, d, seed, normalize=(metric != "l2sq" and kind == 0))
    if kind == 1:
        X = rng.integers(0, 4, size=(n, d)).astype(np.float32)  # ties everywhere, duplicated rows
    elif kind == 2:
        X[rng.integers(0, n, size=20)] = 0
    Q = np.concatenate([X[rng.integers(0, n, size=16)], datagen.mixture(16, d, seed + 1)]).astype(np.float32)
    idx = CpuIndex(orc, d, metric, M, M0, efc, 64, order=1, wave=1)
    idx.reserve(n, 1)
    idx.set_pipeline_check(True)  # the build's level searches (insert mode) are not modelled: only searches count below
    idx.build_batch(np.arange(n), X, int(rng.choice([1, 16, 256])), int(rng.choice([1, 4, 32])))
    base = np.array(idx.pipeline_check_state(), dtype=np.int64)
    for k, ef in ((1, 1), (10, int(rng.integers(1, 65))), (3, int(rng.integers(65, 257))), (100, 256)):
        idx.set_pipeline_check(False)
        want = idx.search_many(Q, k, ef=ef)
        idx.set_pipeline_check(True)
        got = idx.search_many(Q, k, ef=ef)
        for a, b in zip(want, got):
            assert np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8)), (seed, k, ef)
    st = np.array(idx.pipeline_check_state(), dtype=np.int64) - base
    tot += st
    if st[2]:
        print("WRONG PREDICTIONS seed %d: %s  (M %d M0 %d d %d %s efc %d n %d kind %d)" % (seed, st, M, M0, d, metric, efc, n, kind), flush=True)
print("seeds %d..%d: expansions checked %d, left to the plain order (exact tie / NaN) %d, wrong predictions %d" % (
    first, first + count - 1, tot[0], tot[1], tot[2]), flush=True)
sys.exit(1 if tot[2] else 0)
