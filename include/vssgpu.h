/*
 * vssgpu.h — C ABI of the MI355X-native HNSW / array-distance engine (libvssgpu.so).
 *
 * This is the drop-in boundary for the one member the reference's HNSWIndex delegates all of its arithmetic
 * to: `unum::usearch::index_dense_gt<row_t> index` (reference src/include/hnsw/hnsw_index.hpp:45).  Every entry
 * point below names the reference call it replaces.  Plain C: pointers + sizes, int status codes, no
 * exceptions and no torch / HIP types in the signatures (a hipStream_t travels as void*).
 *
 * Status convention: 0 = VSS_OK, anything else = failure with a message in vss_last_error()
 * (mirrors `result.error.what()` of usearch's result structs — reference hnsw_index.cpp:474-476,
 * hnsw_index_physical_create.cpp:188-193).
 *
 * Pointer convention: functions ending in `_device` take DEVICE pointers (already resident in HBM) and are
 * asynchronous on the index's stream; all others take HOST pointers, copy, and return when results are valid.
 *
 * Threading: one handle may be used from any number of host threads.  Searches, exact searches, statistics, vss_timing and
 * vss_save run concurrently (reader lock; each blocking call leases one of eight internal search contexts — the analogue of the
 * usearch context a thread leases, index_dense.hpp:1730-1745); calls that change the index (stage, finalize, add, remove,
 * compact, load, reserve, set_*) are exclusive, as under DuckDB's index lock (hnsw_index.cpp:388, 421, 496).  The explicit
 * contexts 0..3 of vss_search_*_begin / vss_search_batch_end belong to the caller; a mutating call made while one of
 * them has a probe in flight is refused.  vss_last_error() is per calling thread.
 */
#ifndef VSSGPU_H
#define VSSGPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vss_index vss_index;

enum { VSS_OK = 0, VSS_ERROR = 1 };

/* Index metric — reference hnsw_index.cpp:264-268 (METRIC_KIND_MAP: "l2sq" | "cosine" | "ip").
 * Distances are usearch's: l2sq = sum (a-b)^2 (no sqrt), cosine = 1 - cos, ip = 1 - a.b
 * (index_plugins.hpp:977-1053). */
enum { VSS_METRIC_L2SQ = 0, VSS_METRIC_COSINE = 1, VSS_METRIC_IP = 2 };

/* SQL scalar functions over FLOAT[N] — named at reference hnsw_index.cpp:659-673 (DuckDB core v1.4.3):
 * array_distance = sqrt(sum (a-b)^2), array_cosine_distance = 1 - cos, array_negative_inner_product = -a.b */
enum { VSS_FN_ARRAY_DISTANCE = 0, VSS_FN_ARRAY_COSINE_DISTANCE = 1, VSS_FN_ARRAY_NEGATIVE_INNER_PRODUCT = 2 };

/* Tombstone key: std::numeric_limits<row_t>::max() — usearch index.hpp:990, index_dense.hpp:435. */
#define VSS_FREE_KEY INT64_MAX

/* ---- lifetime ------------------------------------------------------------------------------------------- */

/* Replaces index_dense_gt<row_t>::make(metric_punned_t(dim, kind, f32), config) with
 * config.{connectivity=M, connectivity_base=M0, expansion_add=ef_construction, expansion_search=ef_search}
 * — reference hnsw_index.cpp:190-219.  `device` is the HIP device ordinal the index lives on.  M0 < M is refused (the
 * reference overruns its base lists there). */
int vss_create(uint64_t dim, int metric, uint64_t M, uint64_t M0, uint64_t ef_construction, uint64_t ef_search,
               int device, vss_index **out);
/* index.reset() / destructor — reference hnsw_index.cpp:414. */
void vss_destroy(vss_index *index);
/* result.error.what() of the calling thread's last failed call on any handle; valid until its next call. */
const char *vss_last_error(vss_index *index);
/* Run the index's kernels on a caller-owned hipStream_t (NULL = the engine's own stream). */
int vss_set_stream(vss_index *index, void *hip_stream);
/* Block until everything queued on the index's stream has finished. */
int vss_synchronize(vss_index *index);

/* ---- build ---------------------------------------------------------------------------------------------- */

/* index.reserve({members, threads}) — reference hnsw_index_physical_create.cpp:300, hnsw_index.cpp:238,459.
 * Grows device storage; a growing reserve restarts the level generator exactly like usearch replaces its
 * thread contexts (index.hpp:2485,2496). */
int vss_reserve(vss_index *index, uint64_t members, uint64_t threads);

/* Bulk path, step 1 (replaces the Sink/Combine materialisation + the per-row loop of
 * HNSWIndexConstructTask::ExecuteTask, reference hnsw_index_physical_create.cpp:102-110,148-209): copy a
 * chunk of `count` rows (vecs = count x dim contiguous floats, the ARRAY child vector; rowids = row_t[count];
 * validity = DuckDB validity mask words or NULL, a cleared bit means NULL row: skipped exactly like
 * hnsw_index.cpp:467-470) into device storage, draw each row's level, leave it unlinked. */
int vss_stage_batch(vss_index *index, const int64_t *rowids, const float *vecs, const uint64_t *validity,
                    uint64_t count);
int vss_stage_batch_device(vss_index *index, const int64_t *d_rowids, const float *d_vecs, uint64_t count);
/* Bulk path, step 2 (replaces the N concurrent index.add streams scheduled by
 * HNSWIndexConstructionEvent::Schedule, reference hnsw_index_physical_create.cpp:235-247): link every staged
 * row into the graph with the batch-synchronous GPU build. */
int vss_build_finalize(vss_index *index);
/* Incremental path: index.add(rowid, vec) for every valid row of a chunk — reference HNSWIndex::Construct,
 * hnsw_index.cpp:463-478.  Equivalent to vss_stage_batch + vss_build_finalize.  Capacity must have been
 * reserved ("Reserve capacity ahead of insertions!" otherwise, as usearch index.hpp:2728-2731). */
int vss_add_batch(vss_index *index, const int64_t *rowids, const float *vecs, const uint64_t *validity,
                  uint64_t count);
/* Tuning of the batch-synchronous build: batch = clamp(nodes / growth_div, 1, max_batch).  No reference counterpart
 * (usearch's parallel build — N add() streams, hnsw_index_physical_create.cpp:235-247 — has no schedule); with
 * max_batch = 1 the build is the reference's sequential add() loop. */
int vss_set_build_params(vss_index *index, uint64_t max_batch, uint64_t growth_div);
/* on != 0: a bulk build into an EMPTY index (vss_build_finalize after staging from scratch) ends with the reordering of
 * vss_compact — the reference's own compaction order, index_gt::compact index.hpp:3405-3494 — so that a freshly built index
 * already has the nodes of a cluster contiguous in HBM.  Off by default (the reference does not reorder on CREATE INDEX
 * either); costs one greedy descent per node (about 1 s per 10M x 768 rows) and changes no answer except the order of rows
 * at exactly equal distance. */
int vss_set_build_reorder(vss_index *index, int on);

/* ---- search --------------------------------------------------------------------------------------------- */

/* Tuning options of the search engine — NO reference counterpart, and NOTHING a DuckDB integration has to call: every option
 * has a default that is the measured best, and results (row ids, distances, counts, work counters) never depend on any of them.
 * They exist for A/B measurements and for the tests that force every engine shape.  (Rounds 2-5 exported one setter per knob;
 * round 6 folded them into this one call — the ABI a maintainer reads is the reference's call sites, not the builder's
 * switches.)  `name`:
 *   search.waves (2..16) / search.walkers (0 = per launch ..8)   wavefronts per workgroup, and how many of them walk a query
 *   search.solo (0 never, 1 automatic, 2 always) / search.solo_max_queries   one single-wave workgroup per query (few queries, narrow rows)
 *   search.team (0/1)            eight-wave teams in the solo shape
 *   search.crew (0/1; 1|16|4|8 = refinements)   the last walker of a workgroup runs its scoring waves behind barriers
 *   search.pipelined (0/1)       accept phase in the shadow of the successor's row loads (exact)
 *   search.wide_lists (0/1)      limits of 257-512 in 12-wave workgroups
 *   search.visited_compact (0/1) / search.visited_lds_log2_max (0 = default, <= 14) / search.visited_cells_per_limit (0 = rule, >= 4)
 *                                / search.retry_in_place (0/1)     where a walker's visited set lives and what happens when it overflows
 *   search.probe_flag_wait (0/1) host-pointer probes of <= 256 queries wait on a pinned flag instead of the stream
 *   search.lookahead (0..8)      one expansion of look-ahead (off: measured slower)
 *   search.gating (0/1)          a launch is issued when its predecessor on the device starts to drain
 * Unknown names and values out of range are refused (VSS_ERROR, vss_last_error says which); the environment variables of the
 * same knobs (tools/README.md) are read once, in vss_create, through the same checks.  DESIGN.md §4.2 describes each mechanism. */
int vss_set_option(vss_index *index, const char *name, int64_t value);

/* index.ef_search(query, k, ef).dump_to(row_ids) — reference HNSWIndex::InitializeScan hnsw_index.cpp:315-341.
 * ef = 0 means the index's ef_search option.  Writes <= k row ids in ascending distance order, returns the
 * count through *out_count. */
int vss_search(vss_index *index, const float *query, uint64_t k, uint64_t ef, int64_t *out_rowids,
               uint64_t *out_count);
/* The batched probe of PhysicalHNSWIndexJoin::Execute (reference hnsw_optimize_join.cpp:111-168): n_queries
 * independent ef_search calls (hnsw_index.cpp:383-397) issued as ONE kernel launch.  queries = n_queries x dim.
 * out_rowids = n_queries x k (unused tail cells = -1), out_distances optional (index metric, may be NULL),
 * out_counts = results per query. */
int vss_search_batch(vss_index *index, const float *queries, uint64_t n_queries, uint64_t k, uint64_t ef,
                     int64_t *out_rowids, float *out_distances, uint32_t *out_counts);
int vss_search_batch_device(vss_index *index, const float *d_queries, uint64_t n_queries, uint64_t k, uint64_t ef,
                            int64_t *d_out_rowids, float *d_out_distances, uint32_t *d_out_counts);
/* index.filtered_search(query, k, predicate) — usearch index_dense.hpp:625-629 with expansion = ef: a row is admitted to
 * the result iff it is live AND bit `rowid` of `allowed` is set (rows with rowid >= n_bits are rejected); rejected rows
 * are still traversed, exactly like tombstones (index.hpp:3986-3992).  This pushes a WHERE predicate INTO the traversal
 * instead of filtering above the scan as the reference does (hnsw_optimize_scan.cpp:168-198, which can return fewer than
 * k rows) — SURVEY §8f rank 3.  `allowed` = (n_bits + 63) / 64 words. */
int vss_search_batch_filtered(vss_index *index, const float *queries, uint64_t n_queries, uint64_t k, uint64_t ef,
                              const uint64_t *allowed, uint64_t n_bits, int64_t *out_rowids, float *out_distances,
                              uint32_t *out_counts);
int vss_search_batch_filtered_device(vss_index *index, const float *d_queries, uint64_t n_queries, uint64_t k, uint64_t ef,
                                     const uint64_t *d_allowed, uint64_t n_bits, int64_t *d_out_rowids,
                                     float *d_out_distances, uint32_t *d_out_counts);
/* Pipelined form of vss_search_batch_device: `context` (0..3) names one of the index's independent search contexts —
 * the analogue of usearch's per-thread contexts (index.hpp:2213-2240) that let several ef_search calls run
 * concurrently under the reference's shared lock (hnsw_index.cpp:388).  _begin enqueues the probe on the context's own
 * stream and returns; _end waits for it (and re-runs the rare query whose visited set overflowed).  Outputs are valid
 * after _end.  Context 0 is the one the blocking calls use. */
int vss_search_batch_device_begin(vss_index *index, int context, const float *d_queries, uint64_t n_queries, uint64_t k,
                                  uint64_t ef, int64_t *d_out_rowids, float *d_out_distances, uint32_t *d_out_counts);
/* Several probe batches answered by ONE launch of the search engine: `n_batches` (1..32; 16 until round 5) batches of `n_per_batch`
 * queries each, batch b read from d_queries[b] and answered into d_out_rowids[b] / d_out_distances[b] (entries may be
 * NULL) / d_out_counts[b]; the three tables are host arrays of device pointers, read before the call returns.  Every
 * batch gets exactly the answers vss_search_batch_device would give it.  This is what HNSW_INDEX_JOIN's Execute does
 * with the input chunks it has at hand (hnsw_optimize_join.cpp:111-168 probes them one after the other, each waiting
 * for its slowest query): the engine's walkers take queries from all of them until none is left, so the tail of one
 * batch overlaps the body of the next inside one launch.  Completed by vss_search_batch_end(context); the work
 * counters of vss_last_search_stats then cover all the batches. */
int vss_search_multi_device_begin(vss_index *index, int context, uint64_t n_batches, const float *const *d_queries,
                                  uint64_t n_per_batch, uint64_t k, uint64_t ef, int64_t *const *d_out_rowids,
                                  float *const *d_out_distances, uint32_t *const *d_out_counts);
int vss_search_batch_end(vss_index *index, int context);
/* Pipelining policy of the two _begin calls above (default on; tuning, no reference counterpart, results never depend on it).  A launch of the search engine occupies every compute
 * unit, so a second one issued immediately would wait in its hardware queue with its clock running.  With gating on, _begin
 * returns only once the launch begun before it (on another context of this index, or by another index on the same device —
 * e.g. row-range shards sharing one GPU) has handed out its last query — the
 * moment compute units start to fall idle — or has finished; the tail of one launch still overlaps the body of the next,
 * and a launch's measured duration is execution, not queueing.  vss_set_option(index, "search.gating", 0) = issue immediately
 * (round 1's behaviour). */
/* index.ef_search(query, k, ef, thread, exact=true) — usearch search_exact_ index.hpp:4004-4019: brute force
 * over every live row (MFMA distance tiles + exact re-rank).  Same output layout as vss_search_batch. */
int vss_search_exact_batch(vss_index *index, const float *queries, uint64_t n_queries, uint64_t k,
                           int64_t *out_rowids, float *out_distances, uint32_t *out_counts);
int vss_search_exact_batch_device(vss_index *index, const float *d_queries, uint64_t n_queries, uint64_t k,
                                  int64_t *d_out_rowids, float *d_out_distances, uint32_t *d_out_counts);
/* Work counters of the last vss_search_batch* call, summed over its queries (usearch's
 * search_result_t::computed_distances / visited_members, index.hpp:2566-2571):
 * out[0] = computed distances, out[1] = expanded nodes, out[2] = queries, out[3] = retried queries. */
int vss_last_search_stats(vss_index *index, uint64_t *out4);
/* Kernel timing measured with hipEvents on the index's stream (milliseconds): out[0] = search kernel(s) of the last
 * vss_search_batch* call, out[1] = build phase A kernels, out[2] = build phase B (link) kernels, out[3] = build host
 * wall time, out[4] = build batches, out[5] = build batches re-run with a larger visited set (cumulative since the
 * last reset). */
int vss_timing(vss_index *index, double *out6, int reset);
/* Work done by the bulk build so far (cumulative): out[0] = distances computed by the insert searches and their
 * neighbour selection (usearch add_result_t::computed_distances, index.hpp:2505-2510), out[1] = nodes expanded
 * (visited_members), out[2] = distances computed while repairing reverse links. */
int vss_build_work(vss_index *index, uint64_t *out3);
/* Per-query counters of the last host-pointer vss_search_batch call: n_queries x 2 (distances, expansions). */
int vss_last_search_query_stats(vss_index *index, uint32_t *out, uint64_t n_queries);

/* ---- maintenance ---------------------------------------------------------------------------------------- */

/* index.remove(rowid) for each id — reference HNSWIndex::Delete hnsw_index.cpp:496-512; tombstones the node
 * (key := VSS_FREE_KEY), links stay, and the slot joins the free list: later vss_stage_batch / vss_add_batch rows take
 * tombstoned slots over, in the reference's ring order, through its update() path (index_dense.hpp:1766-1793,
 * index.hpp:2801-2859) before any new slot is appended.  *out_removed = number of ids that were present.  Refused while
 * staged rows are still unlinked (the reference has no such state: every add() links immediately). */
int vss_remove_batch(vss_index *index, const int64_t *rowids, uint64_t count, uint64_t *out_removed);
/* index.compact() — reference HNSWIndex::Compact hnsw_index.cpp:481-494 -> index_dense.hpp:1479-1496 -> index_gt::compact
 * index.hpp:3405-3494.  Two effects, both computed on the device:
 *   1. the reference's reordering: every node's cluster = the node its greedy descent from the entry lands on above
 *      level 0 (search_for_one_), nodes renumbered by (level descending, cluster ascending) — ties by old slot, where the
 *      reference's std::sort is unspecified — and every neighbour slot remapped; nodes of one cluster become contiguous in
 *      HBM, so a search touches a few runs of rows instead of rows scattered over the whole table;
 *   2. the DOCUMENTED pruning (reference README.md:69 "pruning deleted items"), which usearch's compact does NOT do
 *      (DESIGN.md): tombstoned nodes are dropped, links to them removed (no new links are made; the
 *      remaining ones keep their order), the free list is emptied.
 * The entry point stays if it survives (else: the surviving node of the highest level, lowest new slot).  Answers of a
 * search are unchanged except for the order of rows at exactly equal distance.  oracle/hnsw_oracle.cpp
 * compact_reordering() / compact_dropping() are the CPU mirrors (streams compared byte for byte).
 * vss_compact_ex(reorder = 0) only prunes (survivors keep their relative order); vss_compact is vss_compact_ex(reorder = 1).
 * The surviving rows are gathered into a second vector buffer; every new array is built aside and swapped in on success,
 * so a failed call leaves the index as it was.  When the second buffer cannot be allocated the call falls back to pruning
 * with the rows moved down in place — no reordering, reported through *out_reordered (may be NULL) — and only that
 * last-resort row move cannot be undone: an error there leaves an index that must be reloaded. */
int vss_compact(vss_index *index);
int vss_compact_ex(vss_index *index, int reorder, int *out_reordered);

/* index.size() / typed size incl. tombstones / index.capacity() / index.max_level() / index.memory_usage()
 * — reference HNSWIndex::GetStats hnsw_index.cpp:292-306, GetInMemorySize.  vss_size = live rows: linked and not
 * tombstoned, plus every staged row (whether it will be appended or take over a tombstoned slot), so it does not jump at
 * vss_build_finalize; vss_nodes = slots in use, tombstones included. */
uint64_t vss_size(vss_index *index);
uint64_t vss_nodes(vss_index *index);
uint64_t vss_capacity(vss_index *index);
uint64_t vss_max_level(vss_index *index);
uint64_t vss_memory_usage(vss_index *index);
uint64_t vss_dimensions(vss_index *index);
int vss_metric(vss_index *index);
/* index.stats(level) — usearch index.hpp:3010-3027: out = {nodes, edges, max_edges, allocated_bytes}. */
int vss_level_stats(vss_index *index, uint64_t level, uint64_t *out4);
/* Progress of a running vss_build_finalize / vss_add_batch, readable from ANOTHER thread without blocking (the building
 * call holds the index meanwhile) — reference PhysicalCreateHNSWIndex::GetSinkProgress
 * hnsw_index_physical_create.cpp:312-327 (built_count against loaded_count).  *linked = rows of the current / last
 * build linked so far, *total = rows that build links. */
int vss_build_progress(vss_index *index, uint64_t *linked, uint64_t *total);

/* ---- persistence ---------------------------------------------------------------------------------------- */

/* index.save_to_stream(cb) / index.load_from_stream(cb) with the callback shape of reference
 * hnsw_index.cpp:548-551 and :234-235.  The byte stream is usearch 2.12's (index_dense.hpp:811-878), so the
 * reference's LinkedBlock writer/reader and a CPU usearch can consume it unchanged.  Callbacks return
 * non-zero on success. */
typedef int (*vss_write_cb)(void *ctx, const void *data, uint64_t size);
typedef int (*vss_read_cb)(void *ctx, void *data, uint64_t size);
uint64_t vss_serialized_length(vss_index *index);
int vss_save(vss_index *index, vss_write_cb write, void *ctx);
int vss_load(vss_index *index, vss_read_cb read, void *ctx);

/* ---- array_* scalar functions ---------------------------------------------------------------------------- */

/* array_distance / array_cosine_distance / array_negative_inner_product over `rows` FLOAT[dim] values
 * (DuckDB core functions named at reference hnsw_index.cpp:659-673).  a = rows x dim contiguous (the ARRAY child
 * vector); b = rows x dim, or ONE vector of dim floats when b_is_constant != 0.  out = rows floats.
 * Edge contract (DuckDB v1.4.3's source is not in the reference tree, so this is the engine's stated behaviour, tested in
 * tests/test_gpu_parity2.py, parity UNPINNED beyond README values):
 *   array_distance                 sqrt(sum (a-b)^2); NaN if any element is NaN; +inf if the sum overflows or an element is inf
 *   array_negative_inner_product   -(sum a*b); IEEE propagation of NaN / inf
 *   array_cosine_distance          1 - clamp(sum a*b / sqrt(sum a^2 * sum b^2), -1, 1): always within [0, 2] for finite
 *                                  non-zero inputs; NaN when either norm is zero (0/0), when an element is NaN or inf, or
 *                                  when the norm product overflows against an overflowing dot product.  (The index
 *                                  metric `cosine` special-cases zero norms as usearch does — that is a different function.) */
int vss_distance_batch(int fn, const float *a, const float *b, int b_is_constant, uint64_t rows, uint64_t dim,
                       float *out, int device);
int vss_distance_batch_device(int fn, const float *d_a, const float *d_b, int b_is_constant, uint64_t rows,
                              uint64_t dim, float *d_out, void *hip_stream);

/* ---- multi-GPU ------------------------------------------------------------------------------------------ */

/* k-way merge after the all-gather of per-shard results (row-range shards: BASELINE configs[3]; the reference has no
 * sharding, its single-index answer is dump_to's ascending (distance, key) list, hnsw_index.cpp:339, which this
 * reproduces over the union): in_* = n_shards x n_queries x k (ascending per shard, unused cells rowid -1 / +inf),
 * out_* = n_queries x k.  Device pointers, async on hip_stream. */
int vss_merge_topk_device(const float *d_in_distances, const int64_t *d_in_rowids, uint64_t n_shards,
                          uint64_t n_queries, uint64_t k, float *d_out_distances, int64_t *d_out_rowids,
                          uint32_t *d_out_counts, void *hip_stream);

/* The same merge over the PACKED exchange layout — one all-gather per launch of the search engine instead of two per
 * batch: every shard (rank) owns one block of vss_packed_block_bytes(n_queries, k) bytes holding the row ids of all the
 * launch's queries (n_queries x k int64, n_queries = batches x queries per batch, batch after batch) followed by their
 * distances (n_queries x k f32), padded to 16 bytes; `d_packed` = the n_shards blocks back to back (what
 * all_gather_into_tensor leaves), 16-byte aligned.  The engine writes a launch's answers straight into a rank's block when
 * vss_search_multi_device_begin is handed pointers into it.  Same ordering contract as above (hnsw_index.cpp:333-339). */
uint64_t vss_packed_block_bytes(uint64_t n_queries, uint64_t k);
int vss_merge_topk_packed_device(const void *d_packed, uint64_t n_shards, uint64_t n_queries, uint64_t k,
                                 float *d_out_distances, int64_t *d_out_rowids, uint32_t *d_out_counts, void *hip_stream);

/* ---- the exchange step of a sharded probe: RCCL all-gather of the packed per-shard blocks over xGMI ------------------------
 * No reference counterpart (the reference is a single-process, single-index library: SURVEY §8e); north_star's "index sharded
 * across the 8 GPUs of one node (row-range partitions, RCCL all-gather of per-shard top-k over xGMI)".  One vss_comm = one
 * rank's RCCL communicator.  RCCL is dlopen()ed on first use: libvssgpu.so has no link-time dependency on it, and a host that
 * already maps an RCCL (PyTorch) shares that copy.  Per launch of the search engine ONE collective moves every rank's block
 * (vss_packed_block_bytes(n_queries, k) bytes, filled directly by vss_search_multi_device_begin) into `gathered` (n_ranks blocks
 * back to back, rank order) on EVERY rank; vss_merge_topk_packed_device(gathered, n_ranks, ...) then merges on whichever rank
 * wants the answer.  All calls return VSS_OK / VSS_ERROR; vss_exchange_last_error() = the calling thread's last failure. */
typedef struct vss_comm vss_comm;
/* 1 if an RCCL library could be loaded (0: vss_exchange_last_error() says why; callers fall back to peer copies) */
int vss_exchange_available(void);
const char *vss_exchange_last_error(void);
/* one process per GPU (`bench.py --gpus N`, a DuckDB worker per device): rank 0 draws the 128-byte id, hands it to the others
 * out of band, and every rank calls init_rank on its own device (collective: returns when all n_ranks have called) */
int vss_exchange_unique_id(void *id128);
int vss_exchange_init_rank(vss_comm **out, int n_ranks, const void *id128, int rank, int device);
/* ONE process driving n distinct devices (host/sharded_index.hpp — the shape a DuckDB process needs): out[i] = the
 * communicator of devices[i].  Devices must be pairwise distinct (RCCL refuses two ranks on one device). */
int vss_exchange_init_all(vss_comm **out, int n, const int *devices);
/* a communicator the caller already owns (ncclComm_t, e.g. the host framework's): used, never destroyed, by this library */
int vss_exchange_adopt(vss_comm **out, void *nccl_comm, int n_ranks, int rank);
int vss_exchange_ranks(vss_comm *comm);
/* the collective: ncclAllGather(local_block -> gathered) of block_bytes bytes per rank, asynchronous on hip_stream (the
 * caller's side stream: the next launches search meanwhile).  One process holding several communicators brackets the calls
 * of one exchange with group_begin / group_end (ncclGroupStart / ncclGroupEnd). */
int vss_exchange_allgather(vss_comm *comm, const void *d_local_block, void *d_gathered, uint64_t block_bytes, void *hip_stream);
int vss_exchange_group_begin(void);
int vss_exchange_group_end(void);
int vss_exchange_destroy(vss_comm *comm);

/* Library / build identification ("gfx950", engine version). */
const char *vss_version(void);

#ifdef __cplusplus
}
#endif
#endif
