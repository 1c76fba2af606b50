// sharded_harness.cpp — drives vss_host::ShardedHNSWIndex (row-range shards, exchange of the per-shard answers, k-way merge)
// on a one-GPU box by placing every shard on device 0 (peer-copy exchange); on a multi-GPU node pass the device ordinals as
// arguments (pairwise distinct ordinals: one RCCL communicator per device, one all-gather per probe — `./sharded_harness 0`
// runs that path with a single rank on a one-GPU box).
//     ./sharded_harness [device ...]
// Checks: routing at the shard boundaries, merged answers == host-side merge of what each shard returns on its own,
// recall against brute force, deleted rows never come back.  Exit code 0 = every check passed.  Needs a MI355X.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>

#include "sharded_index.hpp"

using namespace vss_host;

#define EXPECT(cond)                                                                                                   \
	do {                                                                                                               \
		if (!(cond)) {                                                                                                 \
			std::fprintf(stderr, "CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);                               \
			return 1;                                                                                                  \
		}                                                                                                              \
	} while (0)

int main(int argc, char **argv) {
	std::vector<int> devices;
	for (int i = 1; i < argc; ++i)
		devices.push_back(std::atoi(argv[i]));
	if (devices.empty())
		devices = {0, 0, 0};
	const idx_t G = devices.size(), n = 6001, dim = 32, nq = 257, k = 10;

	// clustered data (seeded), row ids = table positions
	std::mt19937 rng(20260926);
	std::normal_distribution<float> gauss(0.f, 1.f);
	std::vector<float> centres(64 * dim), vecs(n * dim), queries(nq * dim);
	for (auto &c : centres)
		c = gauss(rng);
	auto draw = [&](float *dst) {
		const size_t c = rng() % 64;
		for (idx_t d = 0; d != dim; ++d)
			dst[d] = centres[c * dim + d] + 0.3f * gauss(rng);
	};
	for (idx_t i = 0; i != n; ++i)
		draw(&vecs[i * dim]);
	for (idx_t i = 0; i != nq; ++i)
		draw(&queries[i * dim]);
	std::vector<row_t> ids(n);
	for (idx_t i = 0; i != n; ++i)
		ids[i] = (row_t)i;

	ShardedHNSWIndex index(dim, {{"metric", OptionValue::String("l2sq")}}, n, devices);
	EXPECT(index.ShardCount() == G);
	// distinct devices -> one RCCL communicator per device and one all-gather per probe; shared devices -> peer copies
	std::printf("exchange: %s\n", index.ExchangeKind());
	// routing: every row has exactly one owner, ranges are contiguous and cover [0, n)
	idx_t covered = 0;
	for (idx_t g = 0; g != G; ++g) {
		auto r = ShardedHNSWIndex::ShardRange(g, G, n);
		EXPECT(r.first == covered && r.second >= r.first);
		covered = r.second;
		if (r.second > r.first) {
			EXPECT(index.OwnerOf((row_t)r.first) == g && index.OwnerOf((row_t)r.second - 1) == g);
		}
	}
	EXPECT(covered == n);

	index.BulkReserve(8);
	for (idx_t c = 0; c < n; c += STANDARD_VECTOR_SIZE) { // chunks straddle the shard boundaries
		const idx_t cnt = std::min<idx_t>(STANDARD_VECTOR_SIZE, n - c);
		index.BulkAppendChunk(vecs.data() + c * dim, ids.data() + c, nullptr, cnt);
	}
	index.BulkFinalize();
	EXPECT(index.Count() == n);

	std::vector<row_t> out(nq * k);
	std::vector<float> out_d(nq * k);
	std::vector<uint32_t> counts(nq);
	index.SearchBatch(queries.data(), nq, k, 128, out.data(), out_d.data(), counts.data());

	// brute force on the host
	auto l2 = [&](const float *a, const float *b) {
		float s = 0;
		for (idx_t d = 0; d != dim; ++d)
			s += (a[d] - b[d]) * (a[d] - b[d]);
		return s;
	};
	idx_t hits = 0;
	for (idx_t q = 0; q != nq; ++q) {
		EXPECT(counts[q] == k);
		std::vector<std::pair<float, row_t>> all(n);
		for (idx_t i = 0; i != n; ++i)
			all[i] = {l2(&queries[q * dim], &vecs[i * dim]), (row_t)i};
		std::partial_sort(all.begin(), all.begin() + k, all.end());
		std::set<row_t> truth;
		for (idx_t j = 0; j != k; ++j)
			truth.insert(all[j].second);
		for (idx_t j = 0; j != k; ++j) {
			hits += truth.count(out[q * k + j]);
			if (j)
				EXPECT(out_d[q * k + j - 1] <= out_d[q * k + j]); // ascending
			const float ref = l2(&queries[q * dim], &vecs[out[q * k + j] * dim]);
			EXPECT(std::fabs(out_d[q * k + j] - ref) <= 1e-5f * std::max(ref, 1e-6f)); // the distances are the real ones
		}
	}
	const double recall = (double)hits / (nq * k);
	std::printf("shards %zu, recall@%zu %.4f\n", (size_t)G, (size_t)k, recall);
	EXPECT(recall >= 0.95);

	// delete rows on both sides of every shard boundary plus the current best answers of query 0
	std::vector<row_t> dead(out.begin(), out.begin() + k);
	for (idx_t g = 1; g != G; ++g) {
		auto r = ShardedHNSWIndex::ShardRange(g, G, n);
		dead.push_back((row_t)r.first - 1), dead.push_back((row_t)r.first);
	}
	std::set<row_t> dead_set(dead.begin(), dead.end());
	EXPECT(index.Delete(dead.data(), dead.size()) == dead_set.size());
	EXPECT(index.Count() == n - dead_set.size());
	index.SearchBatch(queries.data(), nq, k, 128, out.data(), out_d.data(), counts.data());
	for (idx_t i = 0; i != nq * k; ++i)
		EXPECT(!dead_set.count(out[i]));
	index.Compact();
	EXPECT(index.Count() == n - dead_set.size());
	index.SearchBatch(queries.data(), nq, k, 128, out.data(), nullptr, nullptr);
	for (idx_t i = 0; i != nq * k; ++i)
		EXPECT(!dead_set.count(out[i]) && out[i] >= 0 && (idx_t)out[i] < n);
	std::printf("sharded harness ok\n");
	return 0;
}
