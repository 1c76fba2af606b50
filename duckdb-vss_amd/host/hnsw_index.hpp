// hnsw_index.hpp — host-side C++ mirror of the reference's `HNSWIndex` (reference src/include/hnsw/hnsw_index.hpp:30-125,
// src/hnsw/hnsw_index.cpp) written over the C ABI of libvssgpu.so instead of usearch.
//
// It keeps the reference's method names, argument meaning and error strings so that the DuckDB glue (BoundIndex
// overrides, PhysicalCreateHNSWIndex, hnsw_index_scan, PhysicalHNSWIndexJoin, pragmas) keeps calling the same five or six
// entry points.  DuckDB types are replaced by their plain payloads (a DataChunk's ARRAY child vector = `const float*`,
// a row-id Vector = `const row_t*`, a ValidityMask = `const uint64_t*`); INTEGRATION.md shows the one-to-one mapping.
// Nothing here computes anything: all arithmetic is behind `vss_*`.
#pragma once
#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vssgpu.h"

namespace vss_host {

using row_t = int64_t;
using idx_t = uint64_t;
constexpr idx_t STANDARD_VECTOR_SIZE = 2048;

// stand-ins for duckdb::BinderException / InternalException (same message text)
struct BinderException : std::runtime_error {
	using std::runtime_error::runtime_error;
};
struct InternalException : std::runtime_error {
	using std::runtime_error::runtime_error;
};

// an option value as the binder sees it: VARCHAR or INTEGER
struct OptionValue {
	bool is_string = false;
	std::string s;
	int32_t i = 0;
	static OptionValue String(std::string v) {
		OptionValue o;
		o.is_string = true;
		o.s = std::move(v);
		return o;
	}
	static OptionValue Integer(int32_t v) {
		OptionValue o;
		o.i = v;
		return o;
	}
};

inline std::string Lower(std::string s) {
	std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
	return s;
}

struct CILess {
	bool operator()(const std::string &a, const std::string &b) const {
		return Lower(a) < Lower(b);
	}
};
using OptionMap = std::map<std::string, OptionValue, CILess>; // case_insensitive_map_t<Value>

struct HNSWLevelStats { // unum::usearch::index_dense_gt<row_t>::stats_t
	idx_t nodes, edges, max_edges, allocated_bytes;
};
struct HNSWIndexStats { // reference hnsw_index.hpp:19-27
	idx_t max_level;
	idx_t count;
	idx_t capacity;
	idx_t approx_size;
	std::vector<HNSWLevelStats> level_stats;
};

struct IndexScanState {
	virtual ~IndexScanState() = default;
};
struct HNSWIndexScanState : IndexScanState { // reference hnsw_index.cpp:309-313
	idx_t current_row = 0;
	idx_t total_rows = 0;
	std::unique_ptr<row_t[]> row_ids;
};
struct MultiScanState final : IndexScanState { // reference hnsw_index.cpp:358-364
	std::vector<row_t> row_ids;
	size_t ef_search;
	explicit MultiScanState(size_t ef_search_p) : ef_search(ef_search_p) {
	}
};

class HNSWIndex {
public:
	static constexpr const char *TYPE_NAME = "HNSW";

	// CREATE INDEX option validation — reference HNSWIndex::CreatePlan, hnsw_index_plan.cpp:33-80 (same messages;
	// pinned by test/sql/hnsw/hnsw_options.test:11-53)
	static void VerifyOptions(const OptionMap &options) {
		for (auto &option : options) {
			const std::string k = Lower(option.first);
			const OptionValue &v = option.second;
			if (k == "metric") {
				if (!v.is_string)
					throw BinderException("HNSW index 'metric' must be a string");
				if (MetricKind(v.s) < 0)
					throw BinderException("HNSW index 'metric' must be one of: 'cosine', 'ip', 'l2sq'");
			} else if (k == "ef_construction" || k == "ef_search") {
				if (v.is_string)
					throw BinderException("HNSW index '" + k + "' must be an integer");
				if (v.i < 1)
					throw BinderException("HNSW index '" + k + "' must be at least 1");
			} else if (k == "m" || k == "m0") {
				const std::string name = k == "m" ? "M" : "M0";
				if (v.is_string)
					throw BinderException("HNSW index '" + name + "' must be an integer");
				if (v.i < 2)
					throw BinderException("HNSW index '" + name + "' must be at least 2");
			} else {
				throw BinderException("Unknown option for HNSW index: '" + option.first + "'");
			}
		}
	}

	// reference constructor hnsw_index.cpp:151-243: option parsing :181-217, reserve(min(32, estimated_cardinality)) :238
	HNSWIndex(idx_t vector_size, const OptionMap &options, idx_t estimated_cardinality, int device = 0) {
		int metric = VSS_METRIC_L2SQ;
		uint64_t ef_construction = 128, ef_search = 64, m = 16, m0 = 32; // usearch defaults, index.hpp:1282-1298
		auto it = options.find("metric");
		if (it != options.end() && MetricKind(it->second.s) >= 0)
			metric = MetricKind(it->second.s);
		if ((it = options.find("ef_construction")) != options.end())
			ef_construction = it->second.i;
		if ((it = options.find("ef_search")) != options.end())
			ef_search = it->second.i;
		if ((it = options.find("m")) != options.end()) {
			m = it->second.i;
			m0 = m * 2;
		}
		if ((it = options.find("m0")) != options.end())
			m0 = it->second.i;
		if (m0 < m) // not a reference message: the reference accepts this and overruns its base lists
			throw BinderException("HNSW index 'M0' must not be smaller than 'M'");
		if (vss_create(vector_size, metric, m, m0, ef_construction, ef_search, device, &index) != VSS_OK)
			throw InternalException("Failed to create the HNSW index: no MI355X / HIP device available");
		ef_search_option = ef_search;
		Check(vss_reserve(index, std::min<idx_t>(32, estimated_cardinality), 1), "reserve");
		index_size = 0;
	}
	~HNSWIndex() {
		vss_destroy(index);
	}
	HNSWIndex(const HNSWIndex &) = delete;
	HNSWIndex &operator=(const HNSWIndex &) = delete;

	idx_t GetVectorSize() const { // hnsw_index.cpp:245-247
		return vss_dimensions(index);
	}
	std::string GetMetric() const { // hnsw_index.cpp:249-260
		switch (vss_metric(index)) {
		case VSS_METRIC_L2SQ:
			return "l2sq";
		case VSS_METRIC_COSINE:
			return "cosine";
		case VSS_METRIC_IP:
			return "ip";
		default:
			throw InternalException("Unknown metric kind");
		}
	}

	std::unique_ptr<HNSWIndexStats> GetStats() { // hnsw_index.cpp:292-306 (loop bound `i < max_level` included)
		std::unique_lock<std::shared_mutex> lock(rwlock);
		auto result = std::make_unique<HNSWIndexStats>();
		result->max_level = vss_max_level(index);
		result->count = vss_size(index);
		result->capacity = vss_capacity(index);
		result->approx_size = vss_memory_usage(index);
		for (idx_t i = 0; i < result->max_level; i++) {
			uint64_t s[4];
			Check(vss_level_stats(index, i, s), "stats");
			result->level_stats.push_back({s[0], s[1], s[2], s[3]});
		}
		return result;
	}

	// ---- single-query scan (HNSW_INDEX_SCAN) — hnsw_index.cpp:315-356.  `hnsw_ef_search` = the session setting
	// (<= 0 when unset): SET hnsw_ef_search > index option > 64.
	std::unique_ptr<IndexScanState> InitializeScan(const float *query_vector, idx_t limit, int64_t hnsw_ef_search = 0) {
		auto state = std::make_unique<HNSWIndexScanState>();
		const idx_t ef = hnsw_ef_search > 0 ? (idx_t)hnsw_ef_search : ef_search_option;
		std::shared_lock<std::shared_mutex> lock(rwlock);
		state->row_ids = std::unique_ptr<row_t[]>(new row_t[std::max<idx_t>(limit, 1)]);
		uint64_t n = 0;
		Check(vss_search(index, query_vector, limit, ef, state->row_ids.get(), &n), "search");
		state->current_row = 0;
		state->total_rows = n;
		return state;
	}
	idx_t Scan(IndexScanState &state, row_t *result, idx_t result_offset) {
		auto &scan_state = static_cast<HNSWIndexScanState &>(state);
		idx_t count = 0;
		row_t *row_ids = result + result_offset;
		while (count < STANDARD_VECTOR_SIZE && scan_state.current_row < scan_state.total_rows)
			row_ids[count++] = scan_state.row_ids[scan_state.current_row++];
		return count;
	}

	// ---- batched probe (HNSW_INDEX_JOIN) — hnsw_index.cpp:366-408
	std::unique_ptr<IndexScanState> InitializeMultiScan(int64_t hnsw_ef_search = 0) {
		return std::make_unique<MultiScanState>(hnsw_ef_search > 0 ? (size_t)hnsw_ef_search : (size_t)ef_search_option);
	}
	idx_t ExecuteMultiScan(IndexScanState &state_p, const float *query_vector, idx_t limit) {
		return ExecuteMultiScanBatch(state_p, query_vector, 1, limit, nullptr);
	}
	// the whole outer chunk in ONE launch instead of <= 2048/k serial ef_search calls
	// (hnsw_optimize_join.cpp:137-149).  Row ids are appended query after query; per-query counts optional.
	idx_t ExecuteMultiScanBatch(IndexScanState &state_p, const float *queries, idx_t n_queries, idx_t limit,
	                            std::vector<uint32_t> *counts_out) {
		auto &state = static_cast<MultiScanState &>(state_p);
		std::vector<row_t> ids(n_queries * limit);
		std::vector<uint32_t> counts(n_queries);
		{
			std::shared_lock<std::shared_mutex> lock(rwlock);
			Check(vss_search_batch(index, queries, n_queries, limit, state.ef_search, ids.data(), nullptr, counts.data()),
			      "search");
		}
		idx_t total = 0;
		for (idx_t q = 0; q != n_queries; ++q) {
			state.row_ids.insert(state.row_ids.end(), ids.begin() + q * limit, ids.begin() + q * limit + counts[q]);
			total += counts[q];
		}
		if (counts_out)
			*counts_out = counts;
		return total;
	}
	const std::vector<row_t> &GetMultiScanResult(IndexScanState &state) {
		return static_cast<MultiScanState &>(state).row_ids;
	}
	void ResetMultiScan(IndexScanState &state) {
		static_cast<MultiScanState &>(state).row_ids.clear();
	}

	// ---- bulk build: what PhysicalCreateHNSWIndex::Sink / Finalize hand over (hnsw_index_physical_create.cpp:102-110,
	// 287-310).  A NULL vector reaching the build is a hard error there (:181-185).
	void BulkReserve(idx_t rows, idx_t threads) {
		std::unique_lock<std::shared_mutex> lock(rwlock);
		Check(vss_reserve(index, rows, threads), "reserve");
	}
	void BulkAppendChunk(const float *vec_child_data, const row_t *rowid_data, const uint64_t *validity, idx_t count) {
		if (validity)
			for (idx_t i = 0; i != count; ++i)
				if (!(validity[i >> 6] & (1ull << (i & 63))))
					throw InternalException("Invalid data in HNSW index construction: Cannot construct an index with NULL values.");
		std::shared_lock<std::shared_mutex> lock(rwlock);
		Check(vss_stage_batch(index, rowid_data, vec_child_data, nullptr, count), "add to");
		index_size += count;
	}
	// PhysicalCreateHNSWIndex::GetSinkProgress (hnsw_index_physical_create.cpp:312-327): built_count / loaded_count,
	// readable while BulkFinalize runs on another thread
	void GetBuildProgress(idx_t &built_count, idx_t &loaded_count) const {
		uint64_t linked = 0, total = 0;
		vss_build_progress(index, &linked, &total);
		built_count = linked;
		loaded_count = total;
	}
	void BulkFinalize() {
		std::unique_lock<std::shared_mutex> lock(rwlock);
		Check(vss_build_finalize(index), "add to");
		is_dirty = true;
	}

	// ---- HNSWIndex::Construct (Append / Insert) — hnsw_index.cpp:421-479
	void Construct(const float *vec_child_data, const row_t *rowid_data, const uint64_t *validity, idx_t count) {
		is_dirty = true;
		idx_t to_add_count = count;
		if (validity) {
			to_add_count = 0;
			for (idx_t i = 0; i != count; ++i)
				to_add_count += (validity[i >> 6] >> (i & 63)) & 1;
		}
		index_size += to_add_count;
		{
			std::unique_lock<std::shared_mutex> lock(rwlock);
			if (index_size > vss_capacity(index))
				Check(vss_reserve(index, NextPowerOfTwo(index_size), 1), "reserve");
		}
		std::shared_lock<std::shared_mutex> lock(rwlock);
		if (vss_add_batch(index, rowid_data, vec_child_data, validity, count) != VSS_OK)
			throw InternalException(std::string("Failed to add to the HNSW index: ") + vss_last_error(index));
	}

	void Delete(const row_t *row_id_data, idx_t count) { // hnsw_index.cpp:496-512
		is_dirty = true;
		std::unique_lock<std::shared_mutex> lock(rwlock);
		uint64_t removed = 0;
		Check(vss_remove_batch(index, row_id_data, count, &removed), "remove from");
		index_size = vss_size(index);
	}

	void Compact() { // hnsw_index.cpp:481-494
		is_dirty = true;
		std::unique_lock<std::shared_mutex> lock(rwlock);
		if (vss_compact(index) != VSS_OK)
			throw InternalException(std::string("Failed to compact the HNSW index: ") + vss_last_error(index));
		index_size = vss_size(index);
	}

	// ---- persistence: the stream the reference writes through LinkedBlockWriter (hnsw_index.cpp:532-554) and reads
	// back through LinkedBlockReader (:223-236).  `block_payload` = DEFAULT_BLOCK_SIZE - sizeof(validity_t) (:45-53).
	std::vector<std::vector<uint8_t>> PersistToBlocks(idx_t block_payload) {
		std::unique_lock<std::shared_mutex> lock(rwlock);
		struct Writer {
			std::vector<std::vector<uint8_t>> blocks;
			idx_t payload;
		} w {{}, block_payload};
		auto cb = [](void *ctx, const void *data, uint64_t size) -> int {
			auto &w = *static_cast<Writer *>(ctx);
			auto *p = static_cast<const uint8_t *>(data);
			while (size) {
				if (w.blocks.empty() || w.blocks.back().size() == w.payload)
					w.blocks.emplace_back();
				const idx_t take = std::min<idx_t>(size, w.payload - w.blocks.back().size());
				w.blocks.back().insert(w.blocks.back().end(), p, p + take);
				p += take;
				size -= take;
			}
			return 1;
		};
		Check(vss_save(index, cb, &w), "serialize");
		is_dirty = false;
		return std::move(w.blocks);
	}
	void LoadFromBlocks(const std::vector<std::vector<uint8_t>> &blocks) {
		std::unique_lock<std::shared_mutex> lock(rwlock);
		struct Reader {
			const std::vector<std::vector<uint8_t>> *blocks;
			size_t block = 0, off = 0;
		} r {&blocks};
		auto cb = [](void *ctx, void *data, uint64_t size) -> int {
			auto &r = *static_cast<Reader *>(ctx);
			auto *p = static_cast<uint8_t *>(data);
			while (size) {
				if (r.block >= r.blocks->size())
					return 0;
				const auto &b = (*r.blocks)[r.block];
				const size_t take = std::min<size_t>(size, b.size() - r.off);
				std::memcpy(p, b.data() + r.off, take);
				p += take, size -= take, r.off += take;
				if (r.off == b.size())
					r.block++, r.off = 0;
			}
			return 1;
		};
		Check(vss_load(index, cb, &r), "load");
		index_size = vss_size(index);
	}

	bool IsDirty() const {
		return is_dirty;
	}
	void SetDirty() {
		is_dirty = true;
	}
	void SyncSize() {
		index_size = vss_size(index);
	}
	idx_t Count() const {
		return vss_size(index);
	}
	vss_index *Handle() {
		return index;
	}

	static idx_t NextPowerOfTwo(idx_t v) {
		idx_t p = 1;
		while (p < v)
			p <<= 1;
		return p;
	}
	static int MetricKind(const std::string &name) { // METRIC_KIND_MAP, hnsw_index.cpp:264-268
		const std::string n = Lower(name);
		if (n == "l2sq")
			return VSS_METRIC_L2SQ;
		if (n == "cosine")
			return VSS_METRIC_COSINE;
		if (n == "ip")
			return VSS_METRIC_IP;
		return -1;
	}

private:
	void Check(int rc, const char *what) {
		if (rc != VSS_OK)
			throw InternalException(std::string("Failed to ") + what + " the HNSW index: " + vss_last_error(index));
	}

	vss_index *index = nullptr; // replaces `unum::usearch::index_dense_gt<row_t> index` (hnsw_index.hpp:45)
	std::shared_mutex rwlock;   // StorageLock rwlock (hnsw_index.hpp:48)
	idx_t index_size = 0;       // atomic<idx_t> index_size (hnsw_index.hpp:51)
	idx_t ef_search_option = 64;
	bool is_dirty = false;
};

} // namespace vss_host
