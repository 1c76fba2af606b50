/*
 * ref_shim.cpp — C shim over the REFERENCE's own vendored usearch 2.12.0 headers.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_api.h).  This file contains no reference source: it
 * `#include`s the headers where they lie under /root/reference/src/include and instantiates
 * `index_dense_gt<int64_t>` exactly as the reference's HNSWIndex constructor does
 * (src/hnsw/hnsw_index.cpp:190-219: metric_punned_t(dim, kind, f32), enable_key_lookups=false,
 * expansion_add / expansion_search / connectivity / connectivity_base from the WITH options).
 *
 * Built by oracle/Makefile into oracle/_ref/libusearch_ref.so (git-ignored; travels with gpurun).
 * Flags mirror the reference's default build: SIMSIMD off (CMakeLists.txt:11-17), no OpenMP, FP16LIB on
 * (src/include/usearch/duckdb_usearch.hpp:6-8), no -march (so no FMA contraction on x86-64).
 */
#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "usearch/duckdb_usearch.hpp"

#include "oracle_api.h"

using namespace unum::usearch;
using index_t = index_dense_gt<int64_t>;

struct orc_index {
	index_t index;
	std::string err;
};

static metric_kind_t kind_of(int metric) {
	switch (metric) {
	case 0:
		return metric_kind_t::l2sq_k;
	case 1:
		return metric_kind_t::cos_k;
	default:
		return metric_kind_t::ip_k;
	}
}

extern "C" {

orc_index *orc_create(uint64_t dim, int metric, uint64_t M, uint64_t M0, uint64_t efc, uint64_t efs) {
	metric_punned_t m(dim, kind_of(metric), scalar_kind_t::f32_k);
	index_dense_config_t config = {};
	config.enable_key_lookups = false;
	config.expansion_add = efc;
	config.expansion_search = efs;
	config.connectivity = M;
	config.connectivity_base = M0;
	auto *h = new orc_index();
	h->index = index_t::make(m, config);
	return h;
}

void orc_destroy(orc_index *h) {
	delete h;
}

const char *orc_last_error(orc_index *h) {
	return h->err.c_str();
}

int orc_reserve(orc_index *h, uint64_t members, uint64_t threads) {
	return h->index.reserve(index_limits_t(members, threads)) ? 0 : 1;
}

int orc_add(orc_index *h, int64_t key, const float *vec, uint64_t *stats) {
	auto r = h->index.add(key, vec, 0);
	if (!r) {
		h->err = r.error.release();
		return 1;
	}
	if (stats) {
		stats[0] = r.computed_distances;
		stats[1] = r.visited_members;
		stats[2] = r.slot;
	}
	return 0;
}

uint64_t orc_search(orc_index *h, const float *q, uint64_t k, uint64_t ef, int exact, int64_t *keys, float *dists,
                    uint64_t *stats) {
	auto r = h->index.ef_search(q, k, ef, 0, exact != 0);
	if (!r) {
		h->err = r.error.release();
		return 0;
	}
	if (stats) {
		stats[0] = r.computed_distances;
		stats[1] = r.visited_members;
	}
	return r.dump_to(keys, dists);
}

uint64_t orc_search_filtered(orc_index *h, const float *q, uint64_t k, uint64_t ef, const uint64_t *allowed,
                             uint64_t n_bits, int64_t *keys, float *dists, uint64_t *stats) {
	auto predicate = [=](int64_t key) {
		return key >= 0 && (uint64_t)key < n_bits && ((allowed[key >> 6] >> (key & 63)) & 1);
	};
	const std::size_t saved = h->index.expansion_search();
	h->index.change_expansion_search(ef);
	auto r = h->index.filtered_search(q, k, predicate, 0, false);
	h->index.change_expansion_search(saved);
	if (!r) {
		h->err = r.error.release();
		return 0;
	}
	if (stats) {
		stats[0] = r.computed_distances;
		stats[1] = r.visited_members;
	}
	return r.dump_to(keys, dists);
}

uint64_t orc_remove(orc_index *h, int64_t key) {
	auto r = h->index.remove(key);
	if (!r) {
		h->err = r.error.release();
		return 0;
	}
	return r.completed;
}

int orc_compact(orc_index *h) {
	auto r = h->index.compact();
	if (!r) {
		h->err = r.error.release();
		return 1;
	}
	return 0;
}

uint64_t orc_size(orc_index *h) {
	return h->index.size();
}
uint64_t orc_nodes(orc_index *h) {
	return h->index.stats().nodes;
}
uint64_t orc_capacity(orc_index *h) {
	return h->index.capacity();
}
uint64_t orc_max_level(orc_index *h) {
	return h->index.max_level();
}
void orc_level_stats(orc_index *h, uint64_t level, uint64_t *out4) {
	auto s = h->index.stats(level);
	out4[0] = s.nodes;
	out4[1] = s.edges;
	out4[2] = s.max_edges;
	out4[3] = s.allocated_bytes;
}

uint64_t orc_serialized_length(orc_index *h) {
	return h->index.serialized_length();
}

int64_t orc_save(orc_index *h, uint8_t *buf, uint64_t cap) {
	uint64_t off = 0;
	bool overflow = false;
	auto r = h->index.save_to_stream([&](const void *data, size_t size) {
		if (off + size > cap) {
			overflow = true;
			return false;
		}
		std::memcpy(buf + off, data, size);
		off += size;
		return true;
	});
	if (!r || overflow) {
		h->err = overflow ? "buffer too small" : r.error.release();
		return -1;
	}
	return (int64_t)off;
}

int orc_load(orc_index *h, const uint8_t *buf, uint64_t len) {
	uint64_t off = 0;
	auto r = h->index.load_from_stream([&](void *data, size_t size) {
		if (off + size > len)
			return false;
		std::memcpy(data, buf + off, size);
		off += size;
		return true;
	});
	if (!r) {
		h->err = r.error.release();
		return 1;
	}
	return 0;
}

// ---- all-cores variants, for bench.py's cpu_baseline only (reference build only; the restatement is single-threaded).
// The reference's own threading model: searches lease one context per thread (index_dense.hpp:1730-1745); the bulk
// build runs one add() stream per scheduler thread over a shared chunk cursor
// (src/hnsw/hnsw_index_physical_create.cpp:148-209, 239-245).  Both return the elapsed seconds, < 0 on error.
// Work stops being handed out after `max_seconds`; *done = units completed.
double orc_search_mt(orc_index *h, const float *Q, uint64_t nq, uint64_t k, uint64_t ef, uint64_t threads,
                     uint64_t total_queries, double max_seconds, int64_t *out_keys, uint64_t *done) {
	if (!h->index.reserve(index_limits_t(h->index.capacity(), threads)))
		return -1.0;
	const uint64_t dim = h->index.dimensions();
	std::atomic<uint64_t> cursor(0), finished(0);
	std::atomic<int> failed(0);
	auto t0 = std::chrono::steady_clock::now();
	auto expired = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > max_seconds; };
	std::vector<std::thread> pool;
	for (uint64_t t = 0; t != threads; ++t)
		pool.emplace_back([&, t] {
			std::vector<int64_t> keys(k);
			for (;;) {
				const uint64_t i = cursor.fetch_add(1);
				if (i >= total_queries || expired())
					break;
				const uint64_t qi = i % nq;
				auto r = h->index.ef_search(Q + qi * dim, k, ef, t, false);
				if (!r) {
					failed = 1;
					break;
				}
				const uint64_t n = r.dump_to(keys.data());
				if (out_keys && i < nq) {
					for (uint64_t j = 0; j != k; ++j)
						out_keys[qi * k + j] = j < n ? keys[j] : -1;
				}
				finished++;
			}
		});
	for (auto &th : pool)
		th.join();
	const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	if (done)
		*done = finished;
	return failed ? -1.0 : s;
}

double orc_add_mt(orc_index *h, const int64_t *keys, const float *vecs, uint64_t n, uint64_t threads, double max_seconds,
                  uint64_t *done) {
	if (!h->index.reserve(index_limits_t(std::max<uint64_t>(h->index.capacity(), h->index.size() + n), threads)))
		return -1.0;
	const uint64_t dim = h->index.dimensions();
	std::atomic<uint64_t> cursor(0), finished(0);
	std::atomic<int> failed(0);
	const uint64_t chunk = 2048; // STANDARD_VECTOR_SIZE: the unit a construct task grabs
	auto t0 = std::chrono::steady_clock::now();
	auto expired = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > max_seconds; };
	std::vector<std::thread> pool;
	for (uint64_t t = 0; t != threads; ++t)
		pool.emplace_back([&, t] {
			for (;;) {
				const uint64_t c0 = cursor.fetch_add(chunk);
				if (c0 >= n || failed)
					break;
				for (uint64_t i = c0; i < std::min(n, c0 + chunk) && !expired(); ++i) {
					if (!h->index.add(keys[i], vecs + i * dim, t)) {
						failed = 1;
						break;
					}
					finished++;
				}
			}
		});
	for (auto &th : pool)
		th.join();
	const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	if (done)
		*done = finished;
	return failed ? -1.0 : s;
}

float orc_distance(int metric, const float *a, const float *b, uint64_t dim) {
	metric_punned_t m(dim, kind_of(metric), scalar_kind_t::f32_k);
	return m(reinterpret_cast<const byte_t *>(a), reinterpret_cast<const byte_t *>(b));
}
}
