"""Row-range sharding of one index over the GPUs of a node (BASELINE configs[3]).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  Rank r owns rows
[r*N/G, (r+1)*N/G) and builds an independent graph over them — no build-time communication.  A search runs the SAME
query batch on every shard with the same k, all-gathers the per-shard (distance f32, rowid i64)[B x k] results
(12*B*k bytes per rank: 120 KiB at B=1024, k=10 — latency-bound, nowhere near the xGMI link budget) and merges them
with one k-way merge per query.  Row ids are global table positions, so shard-local results are globally meaningful;
distances are the index metric on every shard, hence comparable.
"""
import torch
import torch.distributed as dist


def shard_range(rank, world, n_total):
    """Rows [lo, hi) owned by `rank`: contiguous, disjoint, covering, sizes differ by at most one."""
    return rank * n_total // world, (rank + 1) * n_total // world


def owner_of(rowid, world, n_total):
    """Rank whose shard holds `rowid` (deletes are routed to the owner)."""
    r = min(world - 1, (rowid * world) // max(1, n_total))
    while rowid < shard_range(r, world, n_total)[0]:
        r -= 1
    while rowid >= shard_range(r, world, n_total)[1]:
        r += 1
    return r


class ShardedTopK:
    """all-gather + merge of per-shard top-k.  `merge(gathered_d, gathered_ids, out_d, out_ids)` is the k-way merge:
    on the GPU it is vss_merge_topk_device (see bench.py); the tests pass a torch reference."""

    def __init__(self, n_queries, k, device, merge, group=None):
        self.world = dist.get_world_size(group)
        self.group = group
        self.B, self.k = n_queries, k
        self.gath_d = torch.empty((self.world, n_queries, k), dtype=torch.float32, device=device)
        self.gath_i = torch.empty((self.world, n_queries, k), dtype=torch.int64, device=device)
        self.out_d = torch.empty((n_queries, k), dtype=torch.float32, device=device)
        self.out_i = torch.empty((n_queries, k), dtype=torch.int64, device=device)
        self.merge = merge

    def __call__(self, local_d, local_i):
        """local_*: (B, k) ascending per query, unused cells = (+inf, -1).  Returns merged (B, k) tensors."""
        if dist.get_backend(self.group) == "gloo":  # CPU tests and the single-GPU multi-process smoke test
            dist.all_gather(list(self.gath_d.unbind(0)), local_d.contiguous(), group=self.group)
            dist.all_gather(list(self.gath_i.unbind(0)), local_i.contiguous(), group=self.group)
        else:  # RCCL over xGMI
            dist.all_gather_into_tensor(self.gath_d.view(-1), local_d.contiguous().view(-1), group=self.group)
            dist.all_gather_into_tensor(self.gath_i.view(-1), local_i.contiguous().view(-1), group=self.group)
        self.merge(self.gath_d, self.gath_i, self.out_d, self.out_i)
        return self.out_d, self.out_i


def packed_block_bytes(n_queries, k):
    """Bytes of one rank's block of the packed exchange (= vss_packed_block_bytes): ids, then distances, 16-byte padded."""
    return (n_queries * k * 12 + 15) & ~15


class PackedExchange:
    """ONE collective per launch of the search engine (instead of two per batch): a shard's answers to ALL the batches of
    a launch live in one block — row ids [n_batches x B x k] int64 followed by distances [n_batches x B x k] f32 — which
    the search kernel fills directly (bench.py hands vss_search_multi_device_begin pointers into it).  A rank may hold
    several shards (`n_local`: row-range shards placed on one device; 1 for one shard per GPU): its blocks lie back to
    back, exchange() all-gathers them in one call (12 * n_batches * B * k bytes per shard: 1.9 MiB at 16 x 1024 x 10) and
    merges every query of the launch over all world * n_local shards with one k-way merge call
    (`merge(packed, n_shards, n_queries, k, out_d, out_i)` = vss_merge_topk_packed_device on the GPU; the CPU tests pass a
    torch reference)."""

    def __init__(self, n_batches, n_queries, k, device, merge, n_local=1, group=None, always_collective=False):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        # a one-rank group still runs the collective (bring-up on a one-GPU box: RCCL init + all-gather + merge end to end)
        self.collective = self.world > 1 or (always_collective and dist.is_initialized())
        self.collectives = 0  # all-gathers issued so far
        self.G, self.B, self.k, self.n_local = n_batches, n_queries, k, n_local
        self.nq = n_batches * n_queries
        self.block = packed_block_bytes(self.nq, k)
        self.local = torch.zeros(n_local * self.block, dtype=torch.uint8, device=device)
        self.gathered = self.local if not self.collective else torch.zeros(self.world * n_local * self.block, dtype=torch.uint8,
                                                                           device=device)
        self.out_d = torch.empty((self.nq, k), dtype=torch.float32, device=device)
        self.out_i = torch.empty((self.nq, k), dtype=torch.int64, device=device)
        self.merge = merge

    # where the engine writes batch b's answers of local shard s (views into this rank's blocks)
    def ids(self, b, s=0):
        lo = s * self.block
        return self.local[lo:lo + self.nq * self.k * 8].view(torch.int64).view(self.G, self.B, self.k)[b]

    def dists(self, b, s=0):
        lo = s * self.block + self.nq * self.k * 8
        return self.local[lo:lo + self.nq * self.k * 4].view(torch.float32).view(self.G, self.B, self.k)[b]

    def exchange(self):
        """all-gather this rank's blocks and merge; returns (distances, ids) of shape [G, B, k] (views of internal
        buffers).  After a short last launch the cells of its unused batches are merged too and simply ignored."""
        if self.collective:
            if dist.get_backend(self.group) == "gloo":  # CPU tests and the single-GPU multi-process smoke test
                dist.all_gather(list(self.gathered.view(self.world, -1).unbind(0)), self.local, group=self.group)
            else:  # RCCL over xGMI: one collective for the whole launch
                dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)
            self.collectives += 1
        self.merge(self.gathered, self.world * self.n_local, self.nq, self.k, self.out_d, self.out_i)
        return self.out_d.view(self.G, self.B, self.k), self.out_i.view(self.G, self.B, self.k)


def plan_launches(n_steps, per_launch):
    """Cut `n_steps` probe batches into the fewest launches of at most `per_launch` batches, of (nearly) equal size:
    20 steps at 16 per launch = 10 + 10, not 16 + 4.  Returns [(first batch, one past the last)]."""
    n_l = max(1, (n_steps + per_launch - 1) // per_launch)
    sizes = [n_steps // n_l + (1 if j < n_steps % n_l else 0) for j in range(n_l)]
    out, at = [], 0
    for sz in sizes:
        out.append((at, at + sz))
        at += sz
    return out


def run_pipelined(n_steps, depth, per_launch, n_local, pxs, begin, end, exchange=None, settle=None):
    """The probe loop of one rank (bench.py's timed region; engine-agnostic so that the N > 1 control flow runs on CPU
    with gloo ranks): `n_steps` batches in launches of at most `per_launch`, `depth` launches in flight on contexts
    0 .. depth-1, each context with its own PackedExchange `pxs[c]` (answers + gather / merge buffers).

      begin(c, s, b0, b1, px)  issue the search of batches [b0, b1) on local shard s, context c, answers into px
      end(c, s) -> (kernel ms, distances, expansions)   complete it
      exchange(px)             sharded only: issue ONE exchange for the completed launch (all-gather of the packed blocks +
                               merge; bench.py runs it on a side stream while the next launches search); None otherwise
      settle(px)               the context's previous exchange has retired (called before px is refilled); may be None

    Every rank issues its exchanges in the same order (launch order), which is all a collective needs.
    Returns (kernel ms, distances, expansions, kernel launches)."""
    launches = plan_launches(n_steps, per_launch)
    kms, nd, ne = 0.0, 0, 0
    for j in range(len(launches) + depth):
        c = j % depth
        px = pxs[c]
        if j >= depth and j - depth < len(launches):  # complete the launch issued `depth` launches ago on this context
            for s in range(n_local):
                ms, d, e = end(c, s)
                kms, nd, ne = kms + ms, nd + d, ne + e
            if exchange is not None:
                exchange(px)
        if j < len(launches):
            b0, b1 = launches[j]
            if settle is not None:
                settle(px)
            for s in range(n_local):
                begin(c, s, b0, b1, px)
    return kms, nd, ne, len(launches) * n_local
