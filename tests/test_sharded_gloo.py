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


def _eight_rank_worker(rank, world, port, mode, ret):
    """bench.py's N > 1 control flow with the engine replaced by a brute-force shard: run_pipelined (launch plan, `depth`
    launches in flight, one PackedExchange per context) in both modes — `sharded` (every rank answers every batch on its
    row range, ONE all-gather per launch, merge) and `replicated` (every rank its own batches, no collective)."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("vss_sharded", os.path.join(ROOT, "duckdb-vss_amd", "sharded.py"))
        sharded = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(sharded)
        n_total, dim, B, k, steps, depth, per_launch, n_local = 997, 6, 5, 3, 11, 3, 4, 1
        g = torch.Generator().manual_seed(4242)
        X = torch.randn(n_total, dim, generator=g)          # the same table on every rank
        is_sharded = mode == "sharded"
        lo, hi = sharded.shard_range(rank, world, n_total) if is_sharded else (0, n_total)
        # sharded: every rank sees the same batches; replicated: rank r answers batches r*steps .. of a common stream
        Qs = [torch.randn(B, dim, generator=g) for _ in range(steps * (1 if is_sharded else world))]
        mine = Qs if is_sharded else Qs[rank * steps:(rank + 1) * steps]

        def shard_topk(q):  # brute force over this rank's rows, global row ids
            d = ((q[:, None, :] - X[None, lo:hi, :]) ** 2).sum(-1)
            dd, ii = torch.topk(d, min(k, hi - lo), dim=1, largest=False)
            return dd, ii + lo

        pxs = [sharded.PackedExchange(per_launch, B, k, torch.device("cpu"), _packed_torch_merge, n_local=n_local)
               for _ in range(depth)]
        in_flight, merged, local_answers, order = {}, {}, {}, []

        def begin(c, s, b0, b1, px):
            for i, b in enumerate(range(b0, b1)):
                dd, ii = shard_topk(mine[b])
                px.ids(i, s).fill_(-1)
                px.dists(i, s).fill_(float("inf"))
                px.ids(i, s)[:, :ii.shape[1]] = ii
                px.dists(i, s)[:, :dd.shape[1]] = dd
                local_answers[b] = (dd.clone(), ii.clone())
            in_flight[c] = (b0, b1)

        def end(c, s):
            return 0.5, 10, 1

        def exchange(px):
            c = pxs.index(px)
            md, mi = px.exchange()
            b0, b1 = in_flight[c]
            order.append((b0, b1))
            for i, b in enumerate(range(b0, b1)):
                merged[b] = (md[i].clone(), mi[i].clone())

        kms, nd, ne, n_launches = sharded.run_pipelined(steps, depth, per_launch, n_local, pxs, begin, end,
                                                        exchange if is_sharded else None)
        plan = sharded.plan_launches(steps, per_launch)
        ok = n_launches == len(plan) == 3 and [b - a for a, b in plan] == [4, 4, 3] and (kms, nd, ne) == (1.5, 30, 3)
        if is_sharded:
            ok = ok and order == plan and sum(px.collectives for px in pxs) == len(plan)  # one collective per launch, in launch order
            for b in range(steps):  # merged == brute force over the whole table
                d = ((mine[b][:, None, :] - X[None, :, :]) ** 2).sum(-1)
                dd, ii = torch.topk(d, k, dim=1, largest=False)
                ok = ok and torch.allclose(merged[b][0], dd) and torch.equal(torch.sort(merged[b][1], 1).values, torch.sort(ii, 1).values)
        else:
            ok = ok and sum(px.collectives for px in pxs) == 0 and len(local_answers) == steps
            for b in range(steps):  # every rank answered ITS batches on the whole table
                d = ((mine[b][:, None, :] - X[None, :, :]) ** 2).sum(-1)
                ok = ok and torch.allclose(local_answers[b][0], torch.topk(d, k, dim=1, largest=False).values)
        # what bench.py reduces over the ranks: slowest rank's time (MAX), worst recall (MIN), a common ef (MAX)
        t = torch.tensor([float(rank + 1), 1.0 - 0.01 * rank])
        tmax, tmin = t.clone(), t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        ok = ok and float(tmax[0]) == world and abs(float(tmin[1]) - (1.0 - 0.01 * (world - 1))) < 1e-6
        flags = [None] * world
        dist.all_gather_object(flags, bool(ok))
        if rank == 0:
            ret["ok"] = all(flags)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["sharded", "replicated"])
def test_eight_ranks_dry_run_of_both_multi_gpu_modes(mode):
    """`bench.py --gpus 8 --mode sharded | replicated` as far as it can run without GPUs: 8 gloo ranks, one shard per rank
    (n_local = 1), the pipelined launch loop and the packed exchange of the real driver, a brute-force shard as the engine."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() * 11 + (17 if mode == "sharded" else 29)) % 2000
    mp.spawn(_eight_rank_worker, args=(8, port, mode, ret), nprocs=8, join=True)
    assert ret["ok"]


def test_launch_plan_cuts_steps_into_the_fewest_equal_launches():
    """bench.py's timed region: the steps are cut into the fewest launches of at most `per_launch` batches, of (nearly) equal
    size — the driver's 20 steps at 16 per launch are 10 + 10, not 16 + 4 — covering every step exactly once, in order."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vss_sharded", os.path.join(ROOT, "duckdb-vss_amd", "sharded.py"))
    sharded = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharded)
    assert sharded.plan_launches(20, 16) == [(0, 10), (10, 20)]
    assert sharded.plan_launches(128, 16) == [(16 * i, 16 * i + 16) for i in range(8)]
    # round 5: a launch carries up to 32 batches (bench.py --coalesce default): the driver's 20 steps are ONE launch
    assert sharded.plan_launches(20, 32) == [(0, 20)] and sharded.plan_launches(5, 32) == [(0, 5)]
    assert sharded.plan_launches(128, 32) == [(32 * i, 32 * i + 32) for i in range(4)]
    assert sharded.plan_launches(5, 16) == [(0, 5)] and sharded.plan_launches(1, 1) == [(0, 1)]
    assert sharded.plan_launches(7, 1) == [(i, i + 1) for i in range(7)]
    for n in range(1, 70):
        for g in (1, 2, 3, 10, 16, 32):
            plan = sharded.plan_launches(n, g)
            sizes = [b - a for a, b in plan]
            assert plan[0][0] == 0 and plan[-1][1] == n and all(plan[i][1] == plan[i + 1][0] for i in range(len(plan) - 1))
            assert len(plan) == -(-n // g) and max(sizes) <= g and max(sizes) - min(sizes) <= 1
    # the loop itself, without any exchange: every launch begun once and ended once per local shard, contexts round-robin
    log = []
    out = sharded.run_pipelined(7, 3, 2, 2, [object()] * 3, lambda c, s, b0, b1, px: log.append(("b", c, s, b0, b1)),
                                lambda c, s: (log.append(("e", c, s)), (1.0, 2, 3))[1])
    assert out == (8.0, 16, 24, 8)  # 4 launches x 2 local shards
    begun = [x for x in log if x[0] == "b"]
    assert [(x[3], x[4]) for x in begun[::2]] == [(0, 2), (2, 4), (4, 6), (6, 7)] and [x[1] for x in begun[::2]] == [0, 1, 2, 0]
    assert len([x for x in log if x[0] == "e"]) == 8
    # a context is completed before it is reused
    first_reuse = log.index(("b", 0, 0, 6, 7))
    assert ("e", 0, 0) in log[:first_reuse] and ("e", 0, 1) in log[:first_reuse]
