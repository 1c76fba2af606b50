"""GPU side of tests/test_oracle_golden.py::test_random_option_space_* — collected as
tests/test_gpu_parity2.py::test_option_space_fuzz*; also runnable as a script for wider sweeps on the GPU box:

    python tests/gpu_option_fuzz.py [first_seed=0] [n_seeds=40]

For each seed: random (M, M0 >= M, ef_construction, ef_search, dimension, metric), a batched build in random chunk sizes
with random deletions in between (slot reuse), compared with the oracle in kernel mode (order=1, wave=1) after every
round: graph bytes, then batched searches at random k / ef (ids, distance bits, counters)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gpu_common as gc  # noqa: E402
import datagen  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def describe_graph_difference(gblob, cblob, cpu, metric, first_new, end_new):
    """Every differing list with the owner-to-neighbour distances (oracle wave order), to see which tie went which way."""
    import ctypes
    from oracle_lib import parse_stream
    a, b = parse_stream(gblob), parse_stream(cblob)
    mi = ["l2sq", "cosine", "ip"].index(metric)
    V = b["vectors"]
    shown = 0
    print("  rows added this round: slots of keys %d..%d; entry gpu %d cpu %d" % (first_new, end_new - 1, a["entry"], b["entry"]))
    for s in range(a["rows"]):
        for l in range(len(a["adj"][s])):
            ga, ca = a["adj"][s][l], b["adj"][s][l]
            if np.array_equal(ga, ca):
                continue
            def dist(t):
                return float(cpu.lib.orc_distance_wave(mi, V[s].ctypes.data, V[int(t)].ctypes.data, V.shape[1]))
            print("  slot %d (key %d) level %d:\n    gpu %s\n    cpu %s" % (
                s, b["keys"][s], l, [(int(t), dist(t)) for t in ga], [(int(t), dist(t)) for t in ca]))
            shown += 1
            if shown >= 6:
                return


def run(seed, degenerate=False, verbose=False):
    rng = np.random.default_rng(70_000 + seed)
    M = int(rng.integers(2, 21))
    M0 = int(rng.integers(M, 65))
    efc, efs = int(rng.integers(1, 200)), int(rng.integers(1, 120))
    dim = int(rng.choice([1, 2, 3, 5, 16, 33, 100, 128, 200, 768]))
    metric = ["l2sq", "cosine", "ip"][int(rng.integers(3))]
    max_batch, growth_div = int(rng.choice([1, 16, 256, 4096])), int(rng.choice([1, 4, 32]))
    cfg = dict(seed=seed, M=M, M0=M0, efc=efc, efs=efs, dim=dim, metric=metric, max_batch=max_batch, growth_div=growth_div)
    n = 3000
    X = datagen.mixture(n, dim, seed, normalize=metric != "l2sq")
    Q = datagen.mixture(48, dim, seed + 1, n_clusters=30, normalize=metric != "l2sq")
    if degenerate:  # coarse integer lattice (exact ties everywhere), duplicated rows, all-zero rows and queries
        X = np.rint(datagen.mixture(n, dim, seed) * 1.5).astype(np.float32)
        Q = np.rint(datagen.mixture(48, dim, seed + 1, n_clusters=30) * 1.5).astype(np.float32)
        X[rng.integers(0, n, 40)] = X[rng.integers(0, n, 40)]
        X[rng.integers(0, n, 25)] = 0
        Q[:3] = 0
        Q[3:8] = X[rng.integers(0, n, 5)]
    cpu = gc.oracle_index(dim, metric, M, M0, efc)
    gpu = gc.gpu_index(dim, metric, M, M0, efc, efs)
    cpu.reserve(n), gpu.reserve(n)
    gpu.set_build_params(max_batch, growth_div)
    alive, key = [], 0
    for round_ in range(8):
        m = int(rng.integers(1, 500))
        if key + m > n:
            break
        keys = np.arange(key, key + m)
        cpu.build_batch(keys, X[key:key + m], max_batch, growth_div)
        gpu.add(keys, X[key:key + m])
        alive += keys.tolist()
        key += m
        gblob, cblob = gpu.save(), cpu.save()
        diff = gc.first_graph_difference(gblob, cblob, ignore_counts=True)
        if diff is not None:
            if verbose:
                describe_graph_difference(gblob, cblob, cpu, metric, key - m, key)
            return ("graph", round_, cfg, diff)
        kk, ef = int(rng.choice([1, 3, 10, 50])), int(rng.choice([1, 8, 30, 100, 300]))
        gk, gd, gcnt = gpu.search_batch(Q, kk, ef)
        ck, cd, ccnt, cst = cpu.search_many(Q, kk, ef=ef)
        if not (np.array_equal(gk, ck) and np.array_equal(bits(gd), bits(cd)) and np.array_equal(gcnt, ccnt)):
            if verbose:
                gst = gpu.last_query_stats(len(Q))
                for i in range(len(Q)):
                    if not (np.array_equal(gk[i], ck[i]) and np.array_equal(bits(gd[i]), bits(cd[i])) and gcnt[i] == ccnt[i]):
                        print("  query %d: gpu keys %s d %s cnt %d stats %s | cpu keys %s d %s cnt %d stats %s" % (
                            i, gk[i].tolist(), gd[i].tolist(), gcnt[i], gst[i].tolist(), ck[i].tolist(), cd[i].tolist(),
                            ccnt[i], cst[i].tolist()))
            return ("search", round_, cfg, kk, ef)
        if not np.array_equal(gpu.last_query_stats(len(Q)), cst.astype(np.uint32)):
            return ("search counters", round_, cfg, kk, ef)
        dead = [alive.pop(int(rng.integers(len(alive)))) for _ in range(min(len(alive) - 1, int(rng.integers(0, 120))))]
        if dead:
            gpu.remove(np.asarray(dead, dtype=np.int64))
            for k in dead:
                cpu.remove(int(k))
    return None


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    bad = 0
    degenerate = len(sys.argv) > 3 and sys.argv[3] == "degenerate"
    for s in range(first, first + count):
        r = run(s, degenerate=degenerate, verbose=True)
        if r:
            bad += 1
            print("MISMATCH", r, flush=True)
    print("option fuzz: %d seeds, %d mismatches" % (count, bad))
    sys.exit(1 if bad else 0)
