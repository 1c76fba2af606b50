// host_harness.cpp — drives vss_host::HNSWIndex exactly like the three DuckDB callers do (2048-row chunks with validity
// masks and int64 row ids; single query; one outer chunk of batched queries; delete / compact; linked-block
// persistence), on the README data set of the reference (README.md:12-41, test/sql/hnsw/hnsw_result.test).
// Exit code 0 = every check passed.  Needs a MI355X.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <thread>

#include "hnsw_index.hpp"

using namespace vss_host;

#define EXPECT(cond)                                                                                                   \
	do {                                                                                                               \
		if (!(cond)) {                                                                                                 \
			std::fprintf(stderr, "CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);                               \
			return 1;                                                                                                  \
		}                                                                                                              \
	} while (0)

static bool Throws(const OptionMap &o, const char *msg) {
	try {
		HNSWIndex::VerifyOptions(o);
	} catch (const BinderException &e) {
		return std::string(e.what()) == msg;
	}
	return false;
}

int main() {
	// option validation strings pinned by test/sql/hnsw/hnsw_options.test
	EXPECT(Throws({{"ef_construction", OptionValue::Integer(0)}}, "HNSW index 'ef_construction' must be at least 1"));
	EXPECT(Throws({{"ef_search", OptionValue::String("x")}}, "HNSW index 'ef_search' must be an integer"));
	EXPECT(Throws({{"M", OptionValue::Integer(1)}}, "HNSW index 'M' must be at least 2"));
	EXPECT(Throws({{"m0", OptionValue::Integer(1)}}, "HNSW index 'M0' must be at least 2"));
	EXPECT(Throws({{"metric", OptionValue::String("manhattan")}}, "HNSW index 'metric' must be one of: 'cosine', 'ip', 'l2sq'"));
	EXPECT(Throws({{"foo", OptionValue::Integer(1)}}, "Unknown option for HNSW index: 'foo'"));
	HNSWIndex::VerifyOptions({{"ef_construction", OptionValue::Integer(100)}, {"ef_search", OptionValue::Integer(100)},
	                          {"M", OptionValue::Integer(3)}, {"M0", OptionValue::Integer(3)}});

	// README data: all [a,b,c], a,b,c in 1..9
	std::vector<float> vecs;
	std::vector<row_t> ids;
	for (int a = 1; a <= 9; ++a)
		for (int b = 1; b <= 9; ++b)
			for (int c = 1; c <= 9; ++c) {
				vecs.insert(vecs.end(), {(float)a, (float)b, (float)c});
				ids.push_back((row_t)ids.size());
			}
	const idx_t n = ids.size();

	HNSWIndex index(3, {{"metric", OptionValue::String("l2sq")}}, n);
	EXPECT(index.GetVectorSize() == 3 && index.GetMetric() == "l2sq");
	// bulk build as PhysicalCreateHNSWIndex does: reserve, chunks of <= 2048 rows, finalize
	index.BulkReserve(n, 8);
	for (idx_t c = 0; c < n; c += STANDARD_VECTOR_SIZE) {
		const idx_t cnt = std::min<idx_t>(STANDARD_VECTOR_SIZE, n - c);
		index.BulkAppendChunk(vecs.data() + c * 3, ids.data() + c, nullptr, cnt);
	}
	index.BulkFinalize();
	EXPECT(index.Count() == n);

	// HNSW_INDEX_SCAN: SELECT * FROM t ORDER BY array_distance(vec, [1,2,3]) LIMIT 3
	const float q[3] = {1, 2, 3};
	auto st = index.InitializeScan(q, 3);
	row_t out[STANDARD_VECTOR_SIZE];
	EXPECT(index.Scan(*st, out, 0) == 3);
	float dist[3];
	std::vector<float> found;
	for (int i = 0; i < 3; ++i)
		found.insert(found.end(), vecs.begin() + out[i] * 3, vecs.begin() + out[i] * 3 + 3);
	EXPECT(vss_distance_batch(VSS_FN_ARRAY_DISTANCE, found.data(), q, 1, 3, 3, dist, 0) == VSS_OK);
	EXPECT(dist[0] == 0.0f && dist[1] == 1.0f && dist[2] == 1.0f); // hnsw_result.test:23-28
	EXPECT(index.Scan(*st, out, 0) == 0);

	// HNSW_INDEX_JOIN: one outer chunk of queries, k = 2 (hnsw_lateral_join.test shape)
	auto ms = index.InitializeMultiScan(100);
	std::vector<uint32_t> counts;
	EXPECT(index.ExecuteMultiScanBatch(*ms, vecs.data(), 50, 2, &counts) == 100);
	auto &res = index.GetMultiScanResult(*ms);
	for (idx_t i = 0; i != 50; ++i)
		EXPECT(res[2 * i] == (row_t)i); // each stored vector finds itself first
	index.ResetMultiScan(*ms);
	EXPECT(index.ExecuteMultiScan(*ms, q, 3) == 3 && index.GetMultiScanResult(*ms).size() == 3);

	// Append with NULLs, Delete, Compact, stats
	std::vector<float> extra = {0.5f, 0.5f, 0.5f, 10, 10, 10, 20, 20, 20};
	std::vector<row_t> extra_ids = {1000, 1001, 1002};
	uint64_t validity = 0b101; // row 1001 is NULL
	index.Construct(extra.data(), extra_ids.data(), &validity, 3);
	EXPECT(index.Count() == n + 2);
	row_t dead[2] = {0, 1002};
	index.Delete(dead, 2);
	EXPECT(index.Count() == n);
	auto st2 = index.InitializeScan(extra.data(), 1);
	EXPECT(index.Scan(*st2, out, 0) == 1 && out[0] == 1000);
	index.Compact();
	auto stats = index.GetStats();
	EXPECT(stats->count == n && stats->capacity >= n && stats->level_stats.size() == stats->max_level);

	// checkpoint + reload through linked blocks of 256 KiB - 8 bytes
	auto blocks = index.PersistToBlocks(262144 - 8);
	HNSWIndex loaded(3, {}, 0);
	loaded.LoadFromBlocks(blocks);
	EXPECT(loaded.Count() == n);
	auto st3 = loaded.InitializeScan(q, 3);
	row_t out3[STANDARD_VECTOR_SIZE];
	EXPECT(loaded.Scan(*st3, out3, 0) == 3);
	auto st4 = index.InitializeScan(q, 3);
	EXPECT(index.Scan(*st4, out, 0) == 3);
	for (int i = 0; i < 3; ++i)
		EXPECT(out[i] == out3[i]);
	// Concurrent readers (SURVEY §8b "Threading": any number of sessions may scan at once, each leasing a search context —
	// index_dense.hpp:1730-1745): 8 threads issue single-query scans and batched probes at the same time, against the answers
	// taken serially beforehand; a writer (Append) in the middle of it all must neither corrupt nor dead-lock anything.
	{
		const idx_t nq = 96, kq = 5;
		// answers are compared through their distances (the grid is full of equidistant rows; which of them come back may
		// legitimately change once the writer below has added a node and its links)
		auto dists_of = [&](const row_t *o, idx_t c, const float *qv) {
			std::vector<float> d;
			for (idx_t j = 0; j != c; ++j) {
				const float *v = o[j] >= 0 && o[j] < (row_t)n ? vecs.data() + o[j] * 3 : o[j] == 1000 ? extra.data() : nullptr;
				if (!v) {
					d.push_back(-1.f);
					continue;
				}
				float acc = 0;
				for (int x = 0; x < 3; ++x)
					acc += (v[x] - qv[x]) * (v[x] - qv[x]);
				d.push_back(acc);
			}
			return d;
		};
		std::vector<std::vector<float>> serial(nq);
		for (idx_t i = 0; i != nq; ++i) {
			const float *qv = vecs.data() + 3 * (7 * i % n);
			auto s1 = loaded.InitializeScan(qv, kq);
			row_t o[STANDARD_VECTOR_SIZE];
			const idx_t c = loaded.Scan(*s1, o, 0);
			serial[i] = dists_of(o, c, qv);
			// (row 0 was deleted and compacted away above: its own vector is answered by a neighbour at distance <= 1)
			if (!(c == kq && serial[i][0] >= 0.f && serial[i][0] <= 1.f))
				std::fprintf(stderr, "query %llu: %llu results, first id %lld distance %g\n", (unsigned long long)i,
				             (unsigned long long)c, c ? (long long)o[0] : -1ll, c ? serial[i][0] : -1.f);
			EXPECT(c == kq && serial[i][0] >= 0.f && serial[i][0] <= 1.f);
		}
		std::atomic<int> bad {0};
		std::vector<std::thread> pool;
		for (int t = 0; t < 8; ++t)
			pool.emplace_back([&, t] {
				try {
					for (int round = 0; round < 6; ++round) {
						for (idx_t i = t; i < nq; i += 8) { // HNSW_INDEX_SCAN from this session
							const float *qv = vecs.data() + 3 * (7 * i % n);
							auto s1 = loaded.InitializeScan(qv, kq);
							row_t o[STANDARD_VECTOR_SIZE];
							const idx_t c = loaded.Scan(*s1, o, 0);
							if (dists_of(o, c, qv) != serial[i])
								bad++;
						}
						auto m = loaded.InitializeMultiScan(64); // HNSW_INDEX_JOIN chunk from this session
						std::vector<float> qs;
						for (idx_t i = 0; i != 40; ++i)
							qs.insert(qs.end(), vecs.begin() + 3 * (7 * (i + t) % n), vecs.begin() + 3 * (7 * (i + t) % n) + 3);
						std::vector<uint32_t> cnt;
						loaded.ExecuteMultiScanBatch(*m, qs.data(), 40, kq, &cnt);
						auto &r = loaded.GetMultiScanResult(*m);
						idx_t at = 0;
						for (idx_t i = 0; i != 40; ++i) {
							if (dists_of(r.data() + at, cnt[i], qs.data() + 3 * i) != serial[i + t])
								bad++;
							at += cnt[i];
						}
					}
				} catch (const std::exception &e) {
					std::fprintf(stderr, "reader thread %d: %s\n", t, e.what());
					bad++;
				}
			});
		// meanwhile: an Append far away from every query (exclusive inside the engine; readers wait, nobody breaks)
		std::vector<float> far = {500, 500, 500};
		row_t far_id = 5000;
		loaded.Construct(far.data(), &far_id, nullptr, 1);
		for (auto &th : pool)
			th.join();
		EXPECT(bad == 0);
		EXPECT(loaded.Count() == n + 1);
	}
	std::printf("host harness ok\n");
	return 0;
}
