"""N > 1 path on CPU: world_size-2 (and 3) gloo processes exercise the row-range partition, the all-gather layout and
the merge of per-shard top-k.  Per-shard searches are answered by the CPU oracle here (test infrastructure); on the GPU
box the same driver runs over libvssgpu + RCCL (bench.py --gpus N)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _torch_merge(gd, gi, out_d, out_i):
    """Reference k-way merge: ascending by (distance, rowid), invalid cells (rowid < 0) last."""
    G, B, k = gd.shape
    d = gd.permute(1, 0, 2).reshape(B, G * k).clone()
    i = gi.permute(1, 0, 2).reshape(B, G * k)
    d[i < 0] = float("inf")
    key = torch.argsort(i, dim=1, stable=True)
    d2, i2 = torch.gather(d, 1, key), torch.gather(i, 1, key)
    order = torch.argsort(d2, dim=1, stable=True)[:, :k]
    out_d.copy_(torch.gather(d2, 1, order))
    out_i.copy_(torch.gather(i2, 1, order))
    out_i[torch.isinf(out_d)] = -1


def _worker(rank, world, port, n, dim, k, exact, ret):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import datagen
        from oracle_lib import CpuIndex, load_oracle
        import importlib.util
        spec = importlib.util.spec_from_file_location("vss_sharded", os.path.join(ROOT, "duckdb-vss_amd", "sharded.py"))
        sharded = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(sharded)
        X = datagen.mixture(n, dim, 99)
        Q = datagen.mixture(24, dim, 100, n_clusters=int(np.sqrt(n)))
        lo, hi = sharded.shard_range(rank, world, n)
        shard = CpuIndex(load_oracle(), dim, "l2sq", order=1, wave=1)
        shard.reserve(max(1, hi - lo))
        if hi > lo:
            shard.build_batch(np.arange(lo, hi), X[lo:hi], 256, 8)  # keys are GLOBAL row ids
        keys, d, cnt, _ = shard.search_many(Q, k, ef=64, exact=exact)
        topk = sharded.ShardedTopK(len(Q), k, torch.device("cpu"), _torch_merge)
        md, mi = topk(torch.from_numpy(d), torch.from_numpy(keys))
        if rank == 0:
            ret["ids"], ret["d"] = mi.numpy().copy(), md.numpy().copy()
        ranges = [None] * world
        dist.all_gather_object(ranges, (lo, hi))
        if rank == 0:
            ret["ranges"] = ranges
    finally:
        dist.destroy_process_group()


def _run(world, n, dim, k, exact):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() + world * 7 + n) % 2000
    mp.spawn(_worker, args=(world, port, n, dim, k, exact, ret), nprocs=world, join=True)
    return dict(ret)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_exact_topk_equals_global_bruteforce(world, oracle_lib):
    import datagen
    from oracle_lib import CpuIndex
    n, dim, k = 1501, 12, 7
    ret = _run(world, n, dim, k, True)
    assert ret["ranges"][0][0] == 0 and ret["ranges"][-1][1] == n
    assert all(ret["ranges"][i][1] == ret["ranges"][i + 1][0] for i in range(world - 1))  # contiguous, disjoint, covering
    X = datagen.mixture(n, dim, 99)
    Q = datagen.mixture(24, dim, 100, n_clusters=int(np.sqrt(n)))
    whole = CpuIndex(oracle_lib, dim, "l2sq", order=1, wave=1)
    whole.reserve(n)
    whole.build_batch(np.arange(n), X, 256, 8)
    gk, gd, _, _ = whole.search_many(Q, k, exact=True)
    assert np.array_equal(ret["d"].view(np.uint32), gd.view(np.uint32))
    for i in range(len(Q)):
        if len(set(gd[i].tolist())) == k:
            assert np.array_equal(ret["ids"][i], gk[i])


def test_sharded_graph_search_merges_per_shard_results(oracle_lib):
    """Approximate path: the merged list is the k best of the union of the per-shard lists, ascending, no duplicates."""
    n, dim, k = 2000, 16, 10
    ret = _run(2, n, dim, k, False)
    assert np.all(np.diff(ret["d"], axis=1) >= 0)
    for row in ret["ids"]:
        assert len(set(row.tolist())) == k and row.min() >= 0 and row.max() < n


def test_owner_of_rows():
    import importlib.util
    spec = importlib.util.spec_from_file_location("vss_sharded", os.path.join(ROOT, "duckdb-vss_amd", "sharded.py"))
    sharded = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharded)
    for world, n in ((8, 10_000_000), (3, 10), (4, 7), (8, 100_000_001)):
        for r in range(world):
            lo, hi = sharded.shard_range(r, world, n)
            for rowid in {lo, hi - 1, (lo + hi) // 2} if hi > lo else set():
                assert sharded.owner_of(rowid, world, n) == r


def _pipelined_worker(rank, world, port, ret):
    """bench.py's run_steps exchange pattern on CPU: `depth` probes in flight, each with its own gather / merge buffers,
    exchanged in issue order while later batches are already being 'searched'."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("vss_sharded", os.path.join(ROOT, "duckdb-vss_amd", "sharded.py"))
        sharded = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(sharded)
        B, k, depth, steps = 16, 5, 3, 7
        g = torch.Generator().manual_seed(1000 + rank)
        local = []  # this shard's (ascending distances, global row ids) per step
        for s in range(steps):
            d = torch.sort(torch.rand((B, k), generator=g), dim=1).values
            i = (torch.randperm(B * k, generator=g).reshape(B, k) * world + rank).to(torch.int64)
            local.append((d, i))
        mergers = [sharded.ShardedTopK(B, k, torch.device("cpu"), _torch_merge) for _ in range(depth)]
        plain = sharded.ShardedTopK(B, k, torch.device("cpu"), _torch_merge)
        piped = [None] * steps
        for i in range(steps + depth):  # same loop shape as bench.py run_steps
            c = i % depth
            if i >= depth:
                md, mi = mergers[c](*local[i - depth])
                piped[i - depth] = (md.clone(), mi.clone())
        ok = True
        for s in range(steps):
            md, mi = plain(*local[s])
            ok = ok and torch.equal(md, piped[s][0]) and torch.equal(mi, piped[s][1])
            ok = ok and bool(torch.all(md[:, 1:] >= md[:, :-1])) and all(len(set(r.tolist())) == k for r in mi)
        flags = [None] * world
        dist.all_gather_object(flags, ok)
        if rank == 0:
            ret["ok"] = all(flags)
    finally:
        dist.destroy_process_group()


def test_pipelined_exchange_with_per_slot_buffers():
    """Several probes in flight (bench.py --pipeline 3 with --gpus N): every in-flight probe owns its gather and merge
    buffers, ranks issue the exchanges in the same order, and each merged result equals the un-pipelined one."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() * 3 + 11) % 2000
    mp.spawn(_pipelined_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["ok"]


def _packed_torch_merge(packed, n_shards, nq, k, out_d, out_i):
    """Reference merge over the packed layout (what vss_merge_topk_packed_device does on the GPU)."""
    block = packed.numel() // n_shards
    gi = torch.stack([packed[s * block:s * block + nq * k * 8].view(torch.int64).view(nq, k) for s in range(n_shards)])
    gd = torch.stack([packed[s * block + nq * k * 8:s * block + nq * k * 12].view(torch.float32).view(nq, k)
                      for s in range(n_shards)])
    _torch_merge(gd, gi, out_d, out_i)


def _packed_worker(rank, world, port, n_local, ret):
    """One collective per launch: every rank fills its packed block(s) for ALL batches of a launch, one all-gather, one
    merge over all queries; compared with the two-collectives-per-batch ShardedTopK on the same inputs."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("vss_sharded", os.path.join(ROOT, "duckdb-vss_amd", "sharded.py"))
        sharded = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(sharded)
        G, B, k = 3, 7, 5  # nq * k = 105 is odd: the 16-byte padding of the blocks is exercised
        assert sharded.packed_block_bytes(G * B, k) % 16 == 0 and sharded.packed_block_bytes(G * B, k) >= G * B * k * 12
        calls = []

        def counting_merge(*a):
            calls.append(1)
            _packed_torch_merge(*a)

        px = sharded.PackedExchange(G, B, k, torch.device("cpu"), counting_merge, n_local=n_local)
        g = torch.Generator().manual_seed(77 + rank)
        n_shards = world * n_local
        local = {}
        for s in range(n_local):
            for b in range(G):
                d = torch.sort(torch.rand((B, k), generator=g), dim=1).values
                i = (torch.randperm(B * k, generator=g).reshape(B, k) * n_shards + rank * n_local + s).to(torch.int64)
                if (b + s) % 2:  # a shard that found fewer than k rows: unused cells are (+inf, -1)
                    d[:, k - 2:] = float("inf")
                    i[:, k - 2:] = -1
                px.ids(b, s).copy_(i)
                px.dists(b, s).copy_(d)
                local[(s, b)] = (d, i)
        md, mi = px.exchange()
        ok = len(calls) == 1 and md.shape == (G, B, k)
        # reference: gather every (shard, batch) pair the slow way and merge per batch
        for b in range(G):
            mine = torch.stack([local[(s, b)][0] for s in range(n_local)]), torch.stack([local[(s, b)][1] for s in range(n_local)])
            all_d = [torch.zeros_like(mine[0]) for _ in range(world)]
            all_i = [torch.zeros_like(mine[1]) for _ in range(world)]
            dist.all_gather(all_d, mine[0])
            dist.all_gather(all_i, mine[1])
            gd, gi = torch.cat(all_d), torch.cat(all_i)  # [world * n_local, B, k] in global shard order
            od, oi = torch.empty((B, k)), torch.empty((B, k), dtype=torch.int64)
            _torch_merge(gd, gi, od, oi)
            ok = ok and torch.equal(md[b], od) and torch.equal(mi[b], oi)
        flags = [None] * world
        dist.all_gather_object(flags, bool(ok))
        if rank == 0:
            ret["ok"] = all(flags)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_local", [(2, 1), (3, 1), (2, 4)])
def test_packed_exchange_is_one_collective_per_launch(world, n_local):
    """bench.py's N > 1 exchange: the (distance, row id) results of ALL batches of a launch travel in ONE all-gather of the
    packed per-shard blocks and are merged by one call; several shards per rank (co-resident shards) lie back to back."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() * 5 + world * 13 + n_local) % 2000
    mp.spawn(_packed_worker, args=(world, port, n_local, ret), nprocs=world, join=True)
    assert ret["ok"]
