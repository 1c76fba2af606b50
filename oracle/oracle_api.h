/*
 * oracle_api.h — C surface shared by the two CPU checkers under oracle/:
 *
 *   oracle/hnsw_oracle.cpp   the repo's own restatement of the reference algorithm (liboracle.so)
 *   oracle/ref_shim.cpp      a thin shim over the reference's vendored usearch headers, compiled from
 *                            /root/reference where they lie (oracle/_ref/libusearch_ref.so)
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load these libraries, and only as
 * the checker / reported CPU baseline.  The product path (libvssgpu.so) never links or calls them.
 *
 * Both libraries export the same symbols so one ctypes wrapper drives either.
 */
#ifndef VSS_ORACLE_API_H
#define VSS_ORACLE_API_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_index orc_index;

/* metric: 0 = l2sq, 1 = cosine, 2 = ip  (reference: hnsw_index.cpp:264-268 METRIC_KIND_MAP) */
orc_index *orc_create(uint64_t dim, int metric, uint64_t M, uint64_t M0, uint64_t ef_construction, uint64_t ef_search);
void orc_destroy(orc_index *);
const char *orc_last_error(orc_index *);

/* index_dense_gt::reserve(index_limits_t{members, threads}) — index_dense.hpp:753-765 */
int orc_reserve(orc_index *, uint64_t members, uint64_t threads);
/* index_dense_gt::add(key, vec, thread=0) — index_dense.hpp:1748-1794.  stats[0]=computed_distances,
 * stats[1]=visited_members, stats[2]=slot.  Returns 0 on success. */
int orc_add(orc_index *, int64_t key, const float *vec, uint64_t *stats);
/* index_dense_gt::ef_search(q, k, ef, thread=0, exact) — index_dense.hpp:619-623, 1797-1827 */
uint64_t orc_search(orc_index *, const float *q, uint64_t k, uint64_t ef, int exact, int64_t *keys, float *dists,
                    uint64_t *stats);
/* index_dense_gt::filtered_search(q, k, predicate) with expansion_search = ef — index_dense.hpp:625-629, 1821-1826:
 * a member is admitted to the result iff key != free_key && predicate(key).  The predicate is a bitmap over row ids:
 * bit (key) of `allowed` set = admitted; keys >= n_bits are rejected. */
uint64_t orc_search_filtered(orc_index *, const float *q, uint64_t k, uint64_t ef, const uint64_t *allowed,
                             uint64_t n_bits, int64_t *keys, float *dists, uint64_t *stats);
/* index_dense_gt::remove(key) — index_dense.hpp:1228-1255; returns result.completed */
uint64_t orc_remove(orc_index *, int64_t key);
/* index_dense_gt::compact() — index_dense.hpp:1479-1496 */
int orc_compact(orc_index *);

uint64_t orc_size(orc_index *);      /* index_dense_gt::size()  = nodes - free ring */
uint64_t orc_nodes(orc_index *);     /* typed_->size()          = nodes incl. tombstones */
uint64_t orc_capacity(orc_index *);
uint64_t orc_max_level(orc_index *);
/* index_gt::stats(level) — index.hpp:3010-3027 : out = {nodes, edges, max_edges, allocated_bytes} */
void orc_level_stats(orc_index *, uint64_t level, uint64_t *out4);

/* save_to_stream / load_from_stream — index_dense.hpp:811-878, 900-973 */
uint64_t orc_serialized_length(orc_index *);
int64_t orc_save(orc_index *, uint8_t *buf, uint64_t cap); /* bytes written or -1 */
int orc_load(orc_index *, const uint8_t *buf, uint64_t len);

/* metric_{l2sq,cos,ip}_gt<f32> — index_plugins.hpp:977-1053 */
float orc_distance(int metric, const float *a, const float *b, uint64_t dim);

/* liboracle.so only (the reference tree does not hold DuckDB core's array_* functions: PARITY UNPINNED, SURVEY §8c / Appendix
 * B): fn 0 array_distance, 1 array_cosine_distance, 2 array_negative_inner_product; sequential f32 accumulation per row */
void orc_array_function(int fn, const float *a, const float *b, int b_const, uint64_t rows, uint64_t dim, float *out);

#ifdef __cplusplus
}
#endif
#endif
