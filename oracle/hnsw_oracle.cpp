/*
 * hnsw_oracle.cpp — CPU restatement of the reference's HNSW hot path (vendored usearch 2.12.0 as
 * driven by duckdb-vss's HNSWIndex).  Single-threaded, deterministic.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_api.h): tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg are the only callers.  The product (libvssgpu.so) never links this file.
 *
 * PARITY STATUS: pinned.  tests/test_oracle_golden.py checks this file bit-for-bit (serialized graph
 * bytes, result keys, f32 distance bits, computed_distances / visited_members counters) against
 * oracle/_ref/libusearch_ref.so, which is the reference's own usearch headers compiled where they lie;
 * tests/golden/ holds vectors generated from that build so the same checks run where /root/reference is absent.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference/src/include/usearch
 * unless noted).  No reference source text is copied; the algorithms are restated over a plain
 * struct-of-arrays graph instead of usearch's byte tapes, and re-serialised into the reference's stream
 * format only in save()/load().
 *
 * Two switches select what is being restated:
 *   order = 0  "reference order": metrics accumulate sequentially, no FMA (index_plugins.hpp:977-1053 as
 *              compiled for baseline x86-64).  Used for parity against the reference.
 *   order = 1  "wave order": the summation tree of the HIP kernels (duckdb-vss_amd/csrc/wave_primitives.h, wave_distances):
 *              lane g of a G-lane group accumulates float4 chunks g, g+G, ... with fmaf, then an
 *              xor-butterfly over the group.  Used for bit-exact parity of the GPU path.
 *   wave  = 0  candidate handling exactly as the reference (max-heap `next` + sorted `top`).
 *   wave  = 1  candidate handling of the HIP kernels: one sorted list with per-entry "expanded" marks
 *              (plus a second list only when tombstones exist).  Identical results whenever no two
 *              candidate distances tie (proved by tests on tie-free data).
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include "oracle_api.h"

namespace {

constexpr int64_t FREE_KEY = std::numeric_limits<int64_t>::max(); // index.hpp:990, index_dense.hpp:435
constexpr uint32_t FREE_SLOT = 0xFFFFFFFFu;

// ---------------------------------------------------------------------------------------------
// Metrics
// ---------------------------------------------------------------------------------------------

// index_plugins.hpp:1037-1052 / 1006-1026 / 981-993, sequential f32 accumulation.
static float dist_reference_order(int metric, const float *a, const float *b, size_t dim) {
	if (metric == 0) {
		float acc = 0;
		for (size_t i = 0; i != dim; ++i) {
			float d = a[i] - b[i];
			acc += d * d;
		}
		return acc;
	}
	if (metric == 1) {
		float ab = 0, a2 = 0, b2 = 0;
		for (size_t i = 0; i != dim; ++i) {
			float ai = a[i], bi = b[i];
			ab += ai * bi;
			a2 += ai * ai;
			b2 += bi * bi;
		}
		// index_plugins.hpp:1021-1025: [a2==0][b2==0] table
		if (a2 == 0 && b2 == 0)
			return 0;
		if (a2 == 0 || b2 == 0)
			return 1;
		return 1 - ab / (std::sqrt(a2) * std::sqrt(b2));
	}
	float ab = 0;
	for (size_t i = 0; i != dim; ++i)
		ab += a[i] * b[i];
	return 1 - ab;
}

static inline size_t ceil_pow2(size_t v) {
	size_t p = 1;
	while (p < v)
		p <<= 1;
	return p;
}

// The wave summation tree (see header comment; mirrors wave_distance.h).
static void wave_group_geometry(size_t dim, size_t &chunks, size_t &group) {
	chunks = (dim + 3) / 4;
	group = std::min<size_t>(64, ceil_pow2(chunks));
}

static float butterfly(float *lane, size_t group) {
	float tmp[64];
	for (size_t off = group / 2; off >= 1; off /= 2) {
		for (size_t i = 0; i != group; ++i)
			tmp[i] = lane[i] + lane[i ^ off];
		std::memcpy(lane, tmp, group * sizeof(float));
	}
	return lane[0];
}

static float dist_wave_order(int metric, const float *a, const float *b, size_t dim) {
	size_t chunks, group;
	wave_group_geometry(dim, chunks, group);
	float s0[64], s1[64], s2[64];
	for (size_t g = 0; g != group; ++g) {
		float ab = 0, a2 = 0, b2 = 0;
		for (size_t c = g; c < chunks; c += group) {
			for (size_t k = 0; k != 4; ++k) {
				size_t i = 4 * c + k;
				if (i >= dim)
					break;
				float ai = a[i], bi = b[i];
				if (metric == 0) {
					float d = ai - bi;
					ab = fmaf(d, d, ab);
				} else if (metric == 1) {
					ab = fmaf(ai, bi, ab);
					a2 = fmaf(ai, ai, a2);
					b2 = fmaf(bi, bi, b2);
				} else {
					ab = fmaf(ai, bi, ab);
				}
			}
		}
		s0[g] = ab, s1[g] = a2, s2[g] = b2;
	}
	float ab = butterfly(s0, group);
	if (metric == 0)
		return ab;
	if (metric == 2)
		return 1 - ab;
	float a2 = butterfly(s1, group), b2 = butterfly(s2, group);
	if (a2 == 0 && b2 == 0)
		return 0;
	if (a2 == 0 || b2 == 0)
		return 1;
	return 1 - ab / (std::sqrt(a2) * std::sqrt(b2));
}

// ---------------------------------------------------------------------------------------------
// Level generator: std::default_random_engine (= minstd_rand0) + uniform_real_distribution<double>
// as libstdc++ implements them (index.hpp:3723-3727, 2217), restated explicitly.
// ---------------------------------------------------------------------------------------------
struct LevelRng {
	uint64_t x = 1; // default seed
	uint32_t next() {
		x = (x * 16807ull) % 2147483647ull;
		return (uint32_t)x;
	}
	// generate_canonical<double, 53>: k = 2 engine calls, range R = max - min + 1 = 2147483646
	double canonical() {
		const long double R = 2147483646.0L;
		double sum = 0, tmp = 1;
		for (int k = 0; k != 2; ++k) {
			sum += double(next() - 1u) * tmp;
			tmp = (double)((long double)tmp * R);
		}
		double ret = sum / tmp;
		if (ret >= 1.0)
			ret = std::nextafter(1.0, 0.0);
		return ret;
	}
	int16_t level(double inverse_log_connectivity) {
		double u = canonical() * (1.0 - 0.0) + 0.0;
		double r = -std::log(u) * inverse_log_connectivity;
		return (int16_t)r;
	}
};

// ---------------------------------------------------------------------------------------------
// Containers restated from index.hpp
// ---------------------------------------------------------------------------------------------
struct Cand {
	float d;
	uint32_t s;
};

// sorted_buffer_gt — index.hpp:783-917 (ascending; lower_bound insertion: new element before equal ones)
struct SortedTop {
	std::vector<Cand> e;
	size_t lower_bound(float d) const {
		size_t lo = 0, hi = e.size();
		while (lo < hi) {
			size_t mid = (lo + hi) / 2;
			if (e[mid].d < d)
				lo = mid + 1;
			else
				hi = mid;
		}
		return lo;
	}
	void insert_reserved(Cand c) { // :867-875
		e.insert(e.begin() + lower_bound(c.d), c);
	}
	bool insert(Cand c, size_t limit) { // :880-891
		size_t slot = e.empty() ? 0 : lower_bound(c.d);
		if (slot == limit)
			return false;
		if (e.size() == limit)
			e.pop_back();
		e.insert(e.begin() + slot, c);
		return true;
	}
};

// max_heap_gt over {-distance, slot} — index.hpp:620-773; sift order restated literally so that the pop
// order among equal distances matches.
struct NextHeap {
	std::vector<Cand> e; // d holds the NEGATED distance
	static bool less(const Cand &a, const Cand &b) {
		return a.d < b.d;
	}
	void push(Cand c) { // insert_reserved :708-712 + shift_up :752-755
		e.push_back(c);
		size_t i = e.size() - 1;
		while (i && less(e[(i - 1) / 2], e[i])) {
			std::swap(e[(i - 1) / 2], e[i]);
			i = (i - 1) / 2;
		}
	}
	Cand pop() { // :714-721 + shift_down :757-772
		Cand result = e[0];
		std::swap(e[0], e[e.size() - 1]);
		e.pop_back();
		size_t i = 0, n = e.size();
		for (;;) {
			size_t mx = i, l = 2 * i + 1, r = 2 * i + 2;
			if (l < n && less(e[mx], e[l]))
				mx = l;
			if (r < n && less(e[mx], e[r]))
				mx = r;
			if (mx == i)
				break;
			std::swap(e[i], e[mx]);
			i = mx;
		}
		return result;
	}
};

// ring_gt — index.hpp:1150-1277, including the size()==0-when-full quirk.
struct FreeRing {
	std::vector<uint32_t> el;
	size_t cap = 0, head = 0, tail = 0;
	bool empty = true;
	size_t size() const {
		if (empty)
			return 0;
		if (head >= tail)
			return head - tail;
		return cap - (tail - head);
	}
	bool try_pop(uint32_t &v) {
		if (empty)
			return false;
		v = el[tail];
		tail = (tail + 1) % cap;
		empty = head == tail;
		return true;
	}
	void push(uint32_t v) {
		el[head] = v;
		head = (head + 1) % cap;
		empty = false;
	}
	bool reserve(size_t n) {
		if (n < size())
			return false;
		if (n <= cap)
			return true;
		n = std::max<size_t>(ceil_pow2(n), 64);
		std::vector<uint32_t> ne(n);
		size_t i = 0;
		while (try_pop(ne[i]))
			i++;
		el.swap(ne);
		cap = n;
		head = i;
		tail = 0;
		empty = (i == 0);
		return true;
	}
	void clear() {
		head = tail = 0;
		empty = true;
	}
};

// The candidate list of the HIP kernels: ascending by distance, new element before equal ones, bounded,
// each entry carrying an "expanded" mark (duckdb-vss_amd/csrc/wave_list.h).
struct WaveList {
	struct E {
		float d;
		uint32_t s;
		bool expanded;
	};
	std::vector<E> e;
	size_t limit = 0;
	bool insert(float d, uint32_t s) {
		size_t pos = 0;
		while (pos < e.size() && e[pos].d < d)
			pos++;
		if (pos == limit)
			return false;
		if (e.size() == limit)
			e.pop_back();
		e.insert(e.begin() + pos, E {d, s, false});
		return true;
	}
	int first_unexpanded() const {
		for (size_t i = 0; i != e.size(); ++i)
			if (!e[i].expanded)
				return (int)i;
		return -1;
	}
};

// Entries-per-lane of the kernels' register lists: 64*E slots hold a list of `limit` entries.
static size_t wave_list_regs(size_t limit) {
	size_t e = 1;
	while (64 * e < limit)
		e *= 2;
	return e;
}

} // namespace

// ---------------------------------------------------------------------------------------------
// The index
// ---------------------------------------------------------------------------------------------
struct orc_index {
	// configuration — hnsw_index.cpp:181-217 → index_dense_config_t
	size_t dim = 0;
	int metric = 0;
	size_t M = 16, M0 = 32, efc = 128, efs = 64;
	int order = 0;
	int wave = 0;

	// index_gt state — index.hpp:2242-2278
	size_t limit_members = 0, limit_threads = 0;
	size_t capacity = 0;
	size_t count = 0;
	int16_t max_level = -1;
	size_t entry = 0;
	double inv_log_m = 0;
	LevelRng rng;
	uint64_t computed = 0, cycles = 0;

	// node storage (struct of arrays instead of tapes)
	std::vector<int64_t> keys;
	std::vector<int16_t> levels;
	std::vector<std::vector<uint32_t>> lists; // per node: level 0 [cnt, M0 ids], then per level [cnt, M ids]
	std::vector<float> vectors;               // capacity x dim (index_dense vectors_lookup_)

	// index_dense_gt state
	std::unordered_map<int64_t, uint32_t> slot_lookup; // index_dense.hpp:451 (multi=false → one slot per key)
	FreeRing free_keys;                                // index_dense.hpp:463
	size_t tombstones = 0;                             // nodes whose key is FREE_KEY (what the engine tracks)
	std::string err;

	// Model of the engine's register queue (RegQueue, duckdb-vss_amd/csrc/wave_primitives.h) for searches over tombstones /
	// a predicate, 0 = off: at most `regq_cap` candidates wait unexpanded; when one more arrives the farthest one may be
	// forgotten only if the result list is full and it lies beyond the radius, otherwise the query "overflows" (the engine
	// re-runs it with the unbounded queue).  tests/test_oracle_golden.py checks that a query that did not overflow gives
	// exactly the unbounded answer.
	size_t regq_cap = 0;
	bool regq_overflow = false;
	uint64_t regq_drops = 0;
	// Model of the engine's software-pipelined level search (duckdb-vss_amd/csrc/hnsw_kernels.h, level_search_pipelined): the
	// successor of an expansion is told from the fresh scores BEFORE they are inserted.  With the check on, every plain
	// expansion of the kernel-mode search (wave = 1, no tombstones / predicate) predicts its successor by the kernel's rule,
	// performs the sequential inserts, and compares: [0] expansions checked, [1] left to the plain order (an exact tie of the
	// smallest admitted fresh distance, or a NaN), [2] WRONG predictions (must stay 0: tests/test_oracle_golden.py).
	bool pipe_check = false;
	uint64_t pipe_stats[3] = {0, 0, 0};

	// optional result predicate of the running search (filtered_search): bitmap over row ids
	const uint64_t *allowed = nullptr;
	uint64_t allowed_bits = 0;
	bool admitted(size_t slot) const { // index_dense.hpp:1817-1824: key != free_key [&& predicate(key)]
		const int64_t key = keys[slot];
		if (key == FREE_KEY)
			return false;
		if (!allowed)
			return true;
		return key >= 0 && (uint64_t)key < allowed_bits && ((allowed[key >> 6] >> (key & 63)) & 1);
	}

	// scratch
	SortedTop top;
	NextHeap next;
	std::vector<uint8_t> visited_flags;
	std::vector<uint32_t> visited_list;

	float measure(const float *a, const float *b) {
		computed++;
		return order ? dist_wave_order(metric, a, b, dim) : dist_reference_order(metric, a, b, dim);
	}
	const float *vec(size_t slot) const {
		return vectors.data() + slot * dim;
	}
	size_t list_offset(int level) const {
		return level == 0 ? 0 : (M0 + 1) + (size_t)(level - 1) * (M + 1);
	}
	uint32_t *list(size_t slot, int level) {
		return lists[slot].data() + list_offset(level);
	}
	size_t node_bytes(int level) const { // index.hpp:3560-3568
		return 10 + (4 + 4 * M0) + (size_t)level * (4 + 4 * M);
	}

	void visits_clear() {
		for (uint32_t s : visited_list)
			visited_flags[s] = 0;
		visited_list.clear();
		if (visited_flags.size() < capacity)
			visited_flags.resize(capacity, 0);
	}
	bool visits_set(uint32_t s) { // growing_hash_set_gt::set — returns the previous value (index.hpp:1100-1112)
		if (visited_flags[s])
			return true;
		visited_flags[s] = 1;
		visited_list.push_back(s);
		return false;
	}

	// index_gt::reserve — index.hpp:2476-2499 (+ index_dense.hpp:753-765).  A growing reserve replaces every
	// thread context, i.e. restarts the level generator and the counters.
	bool reserve(size_t members, size_t threads) {
		if (threads <= limit_threads && members <= limit_members)
			return true;
		limit_members = members;
		limit_threads = threads;
		capacity = members;
		keys.resize(members, 0);
		levels.resize(members, 0);
		lists.resize(members);
		vectors.resize(members * dim, 0.f);
		rng = LevelRng();
		computed = 0;
		cycles = 0;
		return true;
	}

	// ------------------------------------------------------------------ search_for_one_  index.hpp:3809-3847
	size_t search_for_one(const float *q, size_t closest, int begin_level, int end_level) {
		float closest_dist = measure(q, vec(closest));
		for (int level = begin_level; level > end_level; --level) {
			bool changed;
			do {
				changed = false;
				const uint32_t *nb = list(closest, level);
				uint32_t n = nb[0];
				for (uint32_t i = 0; i != n; ++i) {
					uint32_t cand = nb[1 + i];
					float d = measure(q, vec(cand));
					if (d < closest_dist) {
						closest_dist = d;
						closest = cand;
						changed = true;
					}
				}
				cycles++;
			} while (changed);
		}
		return closest;
	}

	// ------------------------------------------------------------------ search_to_insert_  index.hpp:3855-3921
	void search_to_insert(const float *q, size_t start, size_t new_slot, int level, size_t top_limit) {
		visits_clear();
		next.e.clear();
		top.e.clear();
		float radius = measure(q, vec(start));
		next.push({-radius, (uint32_t)start});
		top.insert_reserved({radius, (uint32_t)start});
		visits_set((uint32_t)start);
		while (!next.e.empty()) {
			Cand c = next.e[0];
			if ((-c.d) > radius && top.e.size() == top_limit)
				break;
			next.pop();
			cycles++;
			if (new_slot == c.s)
				continue;
			const uint32_t *nb = list(c.s, level);
			uint32_t n = nb[0];
			for (uint32_t i = 0; i != n; ++i) {
				uint32_t succ = nb[1 + i];
				if (visits_set(succ))
					continue;
				float d = measure(q, vec(succ));
				if (top.e.size() < top_limit || d < radius) {
					next.push({-d, succ});
					top.insert({d, succ}, top_limit);
					radius = top.e.back().d;
				}
			}
		}
	}

	// ------------------------------------------------------------------ search_to_find_in_base_  index.hpp:3929-3998
	// predicate = "key != free_key" (index_dense.hpp:1815-1820)
	void search_to_find_in_base(const float *q, size_t start, size_t expansion) {
		visits_clear();
		next.e.clear();
		top.e.clear();
		const size_t top_limit = expansion;
		float radius = measure(q, vec(start));
		next.push({-radius, (uint32_t)start});
		visits_set((uint32_t)start);
		if (admitted(start))
			top.insert_reserved({radius, (uint32_t)start});
		while (!next.e.empty()) {
			Cand c = next.e[0];
			// While `top` is still empty the reference's radius is the result of reading an empty buffer (index.hpp:3992,
			// undefined behaviour, in practice garbage that ends the search with no result).  The restatement
			// treats the radius as unbounded until the first admitted entry exists; from then on it is the reference's.
			if (!top.e.empty() && (-c.d) > radius)
				break;
			next.pop();
			cycles++;
			const uint32_t *nb = list(c.s, 0);
			uint32_t n = nb[0];
			for (uint32_t i = 0; i != n; ++i) {
				uint32_t succ = nb[1 + i];
				if (visits_set(succ))
					continue;
				float d = measure(q, vec(succ));
				if (top.e.size() < top_limit || d < radius) {
					next.push({-d, succ});
					if (admitted(succ))
						top.insert({d, succ}, top_limit);
					// index.hpp:3992 reads top.top() even when `top` is empty (undefined behaviour
					// in the reference); the restatement keeps the previous radius in that case.
					if (!top.e.empty())
						radius = top.e.back().d;
				}
			}
		}
	}

	// ------------------------------------------------------------------ the kernels' level search (wave = 1)
	// insert_mode: search_to_insert_ semantics (no tombstone filter, skips expanding new_slot);
	// otherwise search_to_find_in_base_ semantics.  Result left in `top`.
	void wave_level_search(const float *q, size_t start, size_t new_slot, int level, size_t limit, bool insert_mode) {
		visits_clear();
		top.e.clear();
		// the kernel picks its two-list variant when the index holds tombstones at all
		bool any_tomb = !insert_mode && (tombstones > 0 || allowed != nullptr);
		WaveList cand;
		cand.limit = any_tomb ? (size_t)-1 : limit; // tombstones / predicate: every accepted candidate waits in an unbounded queue (the reference's `next` heap)
		SortedTop &res = top;
		float d0 = measure(q, vec(start));
		float radius = d0;
		cand.insert(d0, (uint32_t)start);
		visits_set((uint32_t)start);
		if (any_tomb && admitted(start))
			res.insert_reserved({d0, (uint32_t)start});
		for (;;) {
			int pos = cand.first_unexpanded();
			if (pos < 0)
				break;
			if (any_tomb && !res.e.empty() && cand.e[pos].d > radius)
				break;
			cand.e[pos].expanded = true;
			uint32_t cs = cand.e[pos].s;
			cycles++;
			if (insert_mode && cs == new_slot)
				continue;
			const uint32_t *nb = list(cs, level);
			uint32_t n = nb[0];
			if (pipe_check && !any_tomb && !insert_mode) {
				// the kernel's view of this expansion: the unvisited rows in list order with their distances (one per lane)
				std::vector<std::pair<float, uint32_t>> fresh;
				for (uint32_t i = 0; i != n; ++i) {
					uint32_t succ = nb[1 + i];
					if (!visits_set(succ))
						fresh.push_back({measure(q, vec(succ)), succ});
				}
				// --- the prediction (level_search_pipelined): m = the smallest fresh distance the radius test admits against the
				// radius BEFORE any insert; e = the best unexpanded entry; ties of m (another admitted fresh row, any list entry) or a
				// NaN leave the expansion to the plain order
				const float inf = std::numeric_limits<float>::infinity();
				float m = inf;
				size_t who = 0, at = 0;
				bool nan = false;
				for (size_t i = 0; i != fresh.size(); ++i) {
					const float d = fresh[i].first;
					const bool admitted_now = cand.e.size() < limit || d < radius;
					if (!admitted_now)
						continue;
					nan = nan || !(d == d);
					if (d < m || (who == 0 && d == m))
						m = d, at = i;
				}
				for (size_t i = 0; i != fresh.size(); ++i) {
					const float d = fresh[i].first;
					if ((cand.e.size() < limit || d < radius) && d == m)
						who++;
				}
				bool tie = nan || who > 1;
				if (who >= 1 && !tie)
					for (auto &x : cand.e)
						tie = tie || x.d == m;
				const int e_pos = cand.first_unexpanded();
				uint32_t predicted = 0xFFFFFFFFu;
				if (!tie) {
					if (who >= 1 && (e_pos < 0 || m < cand.e[e_pos].d))
						predicted = fresh[at].second;
					else if (e_pos >= 0)
						predicted = cand.e[e_pos].s;
				}
				// --- the sequential inserts, as always
				for (auto &f : fresh)
					if (cand.e.size() < limit || f.first < radius) {
						cand.insert(f.first, f.second);
						radius = cand.e.back().d;
					}
				// --- and what the plain order picks next
				pipe_stats[0]++;
				if (tie) {
					pipe_stats[1]++;
				} else {
					const int np = cand.first_unexpanded();
					const uint32_t actual = np >= 0 ? cand.e[np].s : 0xFFFFFFFFu;
					if (actual != predicted)
						pipe_stats[2]++;
				}
				continue;
			}
			for (uint32_t i = 0; i != n; ++i) {
				uint32_t succ = nb[1 + i];
				if (visits_set(succ))
					continue;
				float d = measure(q, vec(succ));
				if (!any_tomb) {
					if (cand.e.size() < limit || d < radius) {
						cand.insert(d, succ);
						radius = cand.e.back().d;
					}
				} else {
					if (res.e.size() < limit || d < radius) {
						bool keep = true;
						if (regq_cap && !regq_overflow) {
							size_t pending = 0, last = 0;
							for (size_t j = 0; j != cand.e.size(); ++j)
								if (!cand.e[j].expanded)
									pending++, last = j;
							if (pending >= regq_cap) {
								const bool drop_last = d <= cand.e[last].d; // the new entry goes before equal ones
								const float gone = drop_last ? cand.e[last].d : d;
								if (!(res.e.size() >= limit && gone > radius)) {
									regq_overflow = true;
								} else {
									regq_drops++;
									if (drop_last)
										cand.e.erase(cand.e.begin() + last);
									else
										keep = false;
								}
							}
						}
						if (keep)
							cand.insert(d, succ);
						if (admitted(succ))
							res.insert({d, succ}, limit);
						if (!res.e.empty())
							radius = res.e.back().d;
					}
				}
			}
		}
		if (!any_tomb) {
			res.e.clear();
			for (auto &x : cand.e)
				res.e.push_back({x.d, x.s});
		}
	}

	// ------------------------------------------------------------------ refine_  index.hpp:4027-4063
	size_t refine(size_t needed) {
		std::vector<Cand> &t = top.e;
		size_t top_count = t.size();
		if (top_count < needed)
			return top_count;
		size_t submitted = 1, consumed = 1;
		while (submitted < needed && consumed < top_count) {
			Cand c = t[consumed];
			bool good = true;
			for (size_t i = 0; i < submitted; ++i) {
				float inter = measure(vec(c.s), vec(t[i].s));
				if (inter < c.d) {
					good = false;
					break;
				}
			}
			if (good) {
				t[submitted] = t[consumed];
				submitted++;
			}
			consumed++;
		}
		t.resize(std::min(submitted, t.size()));
		return submitted;
	}

	// ------------------------------------------------------------------ connect_new_node_  index.hpp:3655-3675
	size_t connect_new_node(size_t new_slot, int level) {
		uint32_t *nb = list(new_slot, level);
		size_t n = refine(M);
		for (size_t i = 0; i != n; ++i) {
			nb[1 + nb[0]] = top.e[i].s;
			nb[0]++;
		}
		return nb[1];
	}

	// one reverse link: the body of the loop at index.hpp:3688-3720
	void reconnect_one(size_t close_slot, size_t new_slot, const float *value, int level) {
		size_t connectivity_max = level ? M : M0;
		uint32_t *hdr = list(close_slot, level);
		if (hdr[0] < connectivity_max) {
			hdr[1 + hdr[0]] = (uint32_t)new_slot;
			hdr[0]++;
			return;
		}
		top.e.clear();
		top.insert_reserved({measure(value, vec(close_slot)), (uint32_t)new_slot});
		for (uint32_t i = 0; i != hdr[0]; ++i) {
			uint32_t succ = hdr[1 + i];
			top.insert_reserved({measure(vec(close_slot), vec(succ)), succ});
		}
		// neighbors_ref_t::clear zeroes the count and the used cells (index.hpp:2196-2200)
		std::memset(hdr, 0, (1 + hdr[0]) * sizeof(uint32_t));
		size_t n = refine(connectivity_max);
		for (size_t i = 0; i != n; ++i) {
			hdr[1 + hdr[0]] = top.e[i].s;
			hdr[0]++;
		}
	}

	// ------------------------------------------------------------------ reconnect_neighbor_nodes_  index.hpp:3678-3721
	void reconnect_neighbor_nodes(size_t new_slot, const float *value, int level) {
		// the loop iterates the new node's list in place; copy it so rewriting `top` cannot alias
		const uint32_t *nb = list(new_slot, level);
		std::vector<uint32_t> mine(nb + 1, nb + 1 + nb[0]);
		for (uint32_t close_slot : mine) {
			if (close_slot == new_slot)
				continue;
			reconnect_one(close_slot, new_slot, value, level);
		}
	}

	// ------------------------------------------------------------------ connect_node_across_levels_  index.hpp:3635-3652
	void connect_node_across_levels(const float *value, size_t node_slot, size_t entry_slot, int max_lvl,
	                                int target_level, size_t top_limit) {
		size_t closest = search_for_one(value, entry_slot, max_lvl, target_level);
		for (int level = std::min(target_level, max_lvl); level >= 0; --level) {
			if (wave)
				wave_level_search(value, closest, node_slot, level, top_limit, true);
			else
				search_to_insert(value, closest, node_slot, level, top_limit);
			closest = connect_new_node(node_slot, level);
			reconnect_neighbor_nodes(node_slot, value, level);
		}
	}

	// The limit of the insert search is config.expansion itself (index.hpp:3648 passes it as search_to_insert_'s
	// `top_limit`); max(max(M0, M) + 1, expansion) at index.hpp:2712-2713 is only how much `top` RESERVES.  The two differ
	// when ef_construction < max(M0, M) + 1 (golden case mix1k100_cosine_efc8).
	size_t add_top_limit() const {
		return efc;
	}

	void node_make(size_t slot, int64_t key, int16_t level) { // index.hpp:3582-3592
		lists[slot].assign((M0 + 1) + (size_t)level * (M + 1), 0);
		keys[slot] = key;
		levels[slot] = level;
	}

	// ------------------------------------------------------------------ index_gt::add  index.hpp:2693-2774
	int add_new(int64_t key, const float *value, uint64_t *stats) {
		int16_t max_level_copy = max_level;
		size_t entry_copy = entry;
		int16_t target_level = rng.level(inv_log_m);
		size_t new_slot = count++;
		if (new_slot >= capacity) {
			count--;
			err = "Reserve capacity ahead of insertions!";
			return 1;
		}
		node_make(new_slot, key, target_level);
		// on_success callback (index_dense.hpp:1775-1784): key→slot, copy the vector
		slot_lookup.emplace(key, (uint32_t)new_slot);
		std::memcpy(vectors.data() + new_slot * dim, value, dim * sizeof(float));
		if (stats)
			stats[2] = new_slot;
		if (!new_slot) {
			entry = new_slot;
			max_level = target_level;
			if (stats)
				stats[0] = stats[1] = 0;
			return 0;
		}
		uint64_t c0 = computed, v0 = cycles;
		connect_node_across_levels(value, new_slot, entry_copy, max_level_copy, target_level, add_top_limit());
		if (stats) {
			stats[0] = computed - c0;
			stats[1] = cycles - v0;
		}
		if (target_level > max_level_copy) {
			entry = new_slot;
			max_level = target_level;
		}
		return 0;
	}

	// ------------------------------------------------------------------ index_gt::update  index.hpp:2801-2859
	int update(size_t old_slot, int64_t key, const float *value, uint64_t *stats) {
		int16_t node_level = levels[old_slot];
		std::fill(lists[old_slot].begin(), lists[old_slot].end(), 0u);
		if (keys[old_slot] == FREE_KEY)
			tombstones--;
		keys[old_slot] = 0; // the tape is zeroed, then the level restored (:2839-2840)
		uint64_t c0 = computed, v0 = cycles;
		connect_node_across_levels(value, old_slot, entry, max_level, node_level, add_top_limit());
		keys[old_slot] = key;
		if (stats) {
			stats[0] = computed - c0;
			stats[1] = cycles - v0;
			stats[2] = old_slot;
		}
		// callback runs after the reconnect: the OLD vector was still in place while searching (:2857)
		slot_lookup.emplace(key, (uint32_t)old_slot);
		std::memcpy(vectors.data() + old_slot * dim, value, dim * sizeof(float));
		return 0;
	}

	// ------------------------------------------------------------------ index_dense_gt::add_  index_dense.hpp:1748-1794
	int add(int64_t key, const float *value, uint64_t *stats) {
		if (slot_lookup.count(key)) {
			err = "Duplicate keys not allowed in high-level wrappers";
			return 1;
		}
		uint32_t free_slot = FREE_SLOT;
		free_keys.try_pop(free_slot);
		if (free_slot != FREE_SLOT)
			return update(free_slot, key, value, stats);
		return add_new(key, value, stats);
	}

	// ------------------------------------------------------------------ index_gt::search  index.hpp:2876-2930
	size_t search(const float *q, size_t wanted, size_t ef, bool exact, int64_t *out_keys, float *out_d,
	              uint64_t *stats) {
		if (!wanted)
			return 0;
		if (!ef)
			ef = 64; // default_expansion_search, index.hpp:1296-1298
		top.e.clear();
		uint64_t c0 = computed, v0 = cycles;
		if (!count) {
			if (stats)
				stats[0] = stats[1] = 0;
			return 0;
		}
		if (exact) {
			// search_exact_ index.hpp:4004-4019
			for (size_t i = 0; i != count; ++i) {
				if (keys[i] == FREE_KEY)
					continue;
				float d = measure(q, vec(i));
				top.insert({d, (uint32_t)i}, wanted);
			}
		} else {
			size_t expansion = std::max(ef, wanted);
			size_t closest = search_for_one(q, entry, max_level, 0);
			if (wave)
				wave_level_search(q, closest, FREE_SLOT, 0, expansion, false);
			else
				search_to_find_in_base(q, closest, expansion);
		}
		if (top.e.size() > wanted)
			top.e.resize(wanted);
		if (stats) {
			stats[0] = computed - c0;
			stats[1] = cycles - v0;
		}
		for (size_t i = 0; i != top.e.size(); ++i) {
			if (out_keys)
				out_keys[i] = keys[top.e[i].s];
			if (out_d)
				out_d[i] = top.e[i].d;
		}
		return top.e.size();
	}

	// ------------------------------------------------------------------ index_dense_gt::remove  index_dense.hpp:1228-1255
	size_t remove(int64_t key) {
		auto it = slot_lookup.find(key);
		if (it == slot_lookup.end())
			return 0;
		if (!free_keys.reserve(free_keys.size() + 1)) {
			err = "Can't allocate memory for a free-list";
			return 0;
		}
		uint32_t slot = it->second;
		free_keys.push(slot);
		if (keys[slot] != FREE_KEY)
			tombstones++;
		keys[slot] = FREE_KEY;
		slot_lookup.erase(it);
		return 1;
	}

	// ------------------------------------------------------------------ compact  index.hpp:3405-3494 + index_dense.hpp:1479-1496
	// Reorders by (level desc, cluster asc); keeps tombstones; leaves free_keys / slot_lookup untouched.
	void compact() {
		struct SL {
			uint32_t old_slot, cluster;
			int16_t level;
		};
		std::vector<SL> sl(count);
		for (size_t s = 0; s != count; ++s) {
			size_t cluster = search_for_one(vec(s), entry, max_level, 0);
			sl[s] = {(uint32_t)s, (uint32_t)cluster, levels[s]};
		}
		std::sort(sl.begin(), sl.end(), [](const SL &a, const SL &b) {
			return a.level == b.level ? a.cluster < b.cluster : a.level > b.level;
		});
		std::vector<size_t> old_to_new(count);
		for (size_t n = 0; n != count; ++n)
			old_to_new[sl[n].old_slot] = n;
		std::vector<int64_t> nkeys(keys.size());
		std::vector<int16_t> nlevels(levels.size());
		std::vector<std::vector<uint32_t>> nlists(lists.size());
		std::vector<float> nvec(vectors.size());
		for (size_t n = 0; n != count; ++n) {
			size_t o = sl[n].old_slot;
			nkeys[n] = keys[o];
			nlevels[n] = levels[o];
			nlists[n] = lists[o];
			for (int level = 0; level <= levels[o]; ++level) {
				uint32_t *nb = nlists[n].data() + list_offset(level);
				for (uint32_t i = 0; i != nb[0]; ++i)
					nb[1 + i] = (uint32_t)old_to_new[nb[1 + i]];
			}
			std::memcpy(nvec.data() + n * dim, vec(o), dim * sizeof(float));
		}
		keys.swap(nkeys);
		levels.swap(nlevels);
		lists.swap(nlists);
		vectors.swap(nvec);
		entry = old_to_new[entry];
	}

	// ------------------------------------------------------------------ the ENGINE's compaction (not usearch's: DESIGN.md Q3)
	// What `PRAGMA hnsw_compact_index` is documented to do (reference README.md:69) and what vss_compact does on the GPU:
	// tombstoned nodes are dropped, the survivors keep their order and are renumbered densely, links to dropped nodes are
	// removed (the remaining links keep their order), the entry point stays if it survives, otherwise it becomes the
	// surviving node of the highest level (lowest slot among equals); the free list is emptied.  Mirrored here so that the
	// GPU result can be compared byte for byte (tests/test_gpu_parity2.py).
	void compact_dropping() {
		std::vector<uint32_t> remap(count, FREE_SLOT);
		size_t live = 0;
		for (size_t s = 0; s != count; ++s)
			if (keys[s] != FREE_KEY)
				remap[s] = (uint32_t)live++;
		int16_t nml = -1;
		size_t nentry = 0;
		if (count && remap[entry] != FREE_SLOT) {
			nml = max_level;
			nentry = remap[entry];
		} else {
			for (size_t s = 0; s != count; ++s)
				if (remap[s] != FREE_SLOT && levels[s] > nml)
					nml = levels[s], nentry = remap[s];
		}
		for (size_t s = 0; s != count; ++s) {
			if (remap[s] == FREE_SLOT)
				continue;
			const size_t t = remap[s];
			for (int level = 0; level <= levels[s]; ++level) {
				uint32_t *nb = list(s, level);
				uint32_t kept = 0;
				for (uint32_t i = 0; i != nb[0]; ++i)
					if (remap[nb[1 + i]] != FREE_SLOT)
						nb[1 + kept++] = remap[nb[1 + i]];
				for (uint32_t i = kept; i != nb[0]; ++i)
					nb[1 + i] = 0;
				nb[0] = kept;
			}
			if (t != s) {
				keys[t] = keys[s];
				levels[t] = levels[s];
				lists[t].swap(lists[s]);
				std::memmove(vectors.data() + t * dim, vec(s), dim * sizeof(float));
			}
		}
		for (size_t s = live; s != count; ++s) {
			keys[s] = 0;
			levels[s] = 0;
			lists[s].clear();
		}
		count = live;
		max_level = nml;
		entry = nentry;
		tombstones = 0;
		free_keys.clear();
		slot_lookup.clear();
		for (size_t s = 0; s != count; ++s)
			slot_lookup[keys[s]] = (uint32_t)s;
	}

	// ------------------------------------------------------------------ the ENGINE's vss_compact: reference order + pruning
	// index_gt::compact's reordering (index.hpp:3405-3494: cluster = search_for_one_ from the entry down to level 1, sort
	// by (level descending, cluster ascending)) COMBINED with the documented pruning of compact_dropping() above.  Clusters
	// are taken on the graph as it stands (tombstoned nodes are still traversed, as in the reference), only survivors are
	// numbered, ties of (level, cluster) keep ascending old slots (the reference's std::sort leaves them to the library),
	// links to dropped nodes disappear, the others are remapped in place.  New entry: the old one if it survives, else new
	// slot of the highest level with the lowest new slot.
	void compact_reordering() {
		struct SL {
			uint32_t old_slot, cluster;
			int16_t level;
		};
		std::vector<SL> sl;
		for (size_t s = 0; s != count; ++s) {
			if (keys[s] == FREE_KEY)
				continue;
			const size_t cluster = search_for_one(vec(s), entry, max_level, 0);
			sl.push_back({(uint32_t)s, (uint32_t)cluster, levels[s]});
		}
		std::stable_sort(sl.begin(), sl.end(), [](const SL &a, const SL &b) {
			return a.level == b.level ? a.cluster < b.cluster : a.level > b.level;
		});
		const size_t live = sl.size();
		std::vector<uint32_t> remap(count, FREE_SLOT);
		for (size_t n = 0; n != live; ++n)
			remap[sl[n].old_slot] = (uint32_t)n;
		int16_t nml = -1;
		size_t nentry = 0;
		if (count && remap[entry] != FREE_SLOT) {
			nml = max_level;
			nentry = remap[entry];
		} else {
			for (size_t n = 0; n != live; ++n)
				if (sl[n].level > nml)
					nml = sl[n].level, nentry = n;
		}
		std::vector<int64_t> nkeys(keys.size(), 0);
		std::vector<int16_t> nlevels(levels.size(), 0);
		std::vector<std::vector<uint32_t>> nlists(lists.size());
		std::vector<float> nvec(vectors.size(), 0.f);
		for (size_t n = 0; n != live; ++n) {
			const size_t o = sl[n].old_slot;
			nkeys[n] = keys[o];
			nlevels[n] = levels[o];
			nlists[n] = lists[o];
			for (int level = 0; level <= levels[o]; ++level) {
				uint32_t *nb = nlists[n].data() + list_offset(level);
				uint32_t kept = 0;
				for (uint32_t i = 0; i != nb[0]; ++i)
					if (remap[nb[1 + i]] != FREE_SLOT)
						nb[1 + kept++] = remap[nb[1 + i]];
				for (uint32_t i = kept; i != nb[0]; ++i)
					nb[1 + i] = 0;
				nb[0] = kept;
			}
			std::memcpy(nvec.data() + n * dim, vec(o), dim * sizeof(float));
		}
		keys.swap(nkeys);
		levels.swap(nlevels);
		lists.swap(nlists);
		vectors.swap(nvec);
		count = live;
		max_level = nml;
		entry = nentry;
		tombstones = 0;
		free_keys.clear();
		slot_lookup.clear();
		for (size_t s = 0; s != count; ++s)
			slot_lookup[keys[s]] = (uint32_t)s;
	}

	// ------------------------------------------------------------------ stream format
	size_t serialized_length() const { // index_dense.hpp:883-891 + index.hpp:3097-3102
		size_t n = 8 + count * dim * 4 + 64 + 40;
		for (size_t i = 0; i != count; ++i)
			n += node_bytes(levels[i]) + 2;
		return n;
	}

	int64_t save(uint8_t *buf, size_t cap) { // index_dense.hpp:811-878, index.hpp:3107-3147
		if (serialized_length() > cap)
			return -1;
		uint8_t *p = buf;
		auto put = [&](const void *src, size_t n) {
			std::memcpy(p, src, n);
			p += n;
		};
		uint32_t dims[2] = {(uint32_t)count, (uint32_t)(dim * 4)};
		put(dims, 8);
		put(vectors.data(), count * dim * 4);
		uint8_t head[64];
		std::memset(head, 0, 64);
		std::memcpy(head, "usearch", 7);
		uint16_t ver[3] = {2, 12, 0};
		std::memcpy(head + 7, ver, 6);
		head[13] = metric == 0 ? 'e' : metric == 1 ? 'c' : 'i'; // index_plugins.hpp:103-109
		head[14] = 11;                                          // scalar_kind_t::f32_k
		head[15] = 20;                                          // key kind  i64
		head[16] = 15;                                          // slot kind u32
		uint64_t present = count - free_keys.size(), deleted = free_keys.size(), dimensions = dim;
		std::memcpy(head + 17, &present, 8);
		std::memcpy(head + 25, &deleted, 8);
		std::memcpy(head + 33, &dimensions, 8);
		head[41] = 0; // multi
		put(head, 64);
		uint64_t gh[5] = {count, M, M0, (uint64_t)(int64_t)max_level, entry};
		put(gh, 40);
		for (size_t i = 0; i != count; ++i)
			put(&levels[i], 2);
		for (size_t i = 0; i != count; ++i) {
			put(&keys[i], 8);
			put(&levels[i], 2);
			put(lists[i].data(), lists[i].size() * 4);
		}
		return p - buf;
	}

	int load(const uint8_t *buf, size_t len) { // index_dense.hpp:900-973, index.hpp:3153-3205, reindex_keys_ :1901-1929
		const uint8_t *p = buf, *end = buf + len;
		auto get = [&](void *dst, size_t n) {
			if (p + n > end)
				return false;
			std::memcpy(dst, p, n);
			p += n;
			return true;
		};
		uint32_t dims[2];
		if (!get(dims, 8)) {
			err = "Failed to read 32-bit dimensions of the matrix";
			return 1;
		}
		size_t rows = dims[0], cols = dims[1];
		std::vector<float> nvec(rows * cols / 4);
		if (!get(nvec.data(), rows * cols)) {
			err = "Failed to read vectors";
			return 1;
		}
		uint8_t head[64];
		if (!get(head, 64)) {
			err = "Failed to read the index ";
			return 1;
		}
		if (std::memcmp(head, "usearch", 7) != 0) {
			err = "Magic header mismatch - the file isn't an index";
			return 1;
		}
		uint16_t ver_major;
		std::memcpy(&ver_major, head + 7, 2);
		if (ver_major != 2) {
			err = "File format may be different, please rebuild";
			return 1;
		}
		if (head[15] != 20) {
			err = "Key type doesn't match, consider rebuilding";
			return 1;
		}
		if (head[16] != 15) {
			err = "Slot type doesn't match, consider rebuilding";
			return 1;
		}
		uint64_t dimensions;
		std::memcpy(&dimensions, head + 33, 8);
		int m = head[13] == 'e' ? 0 : head[13] == 'c' ? 1 : 2;
		uint64_t gh[5];
		if (!get(gh, 40)) {
			err = "Failed to pull the header from the stream";
			return 1;
		}
		// reset + adopt
		*this = orc_index_with(dimensions, m, gh[1], gh[2], efc, efs, order, wave);
		if (!gh[0]) {
			if (rows) {
				err = "Index size and the number of vectors doesn't match";
				return 1;
			}
			return 0;
		}
		std::vector<int16_t> lv(gh[0]);
		if (!get(lv.data(), gh[0] * 2)) {
			err = "Failed to pull nodes levels from the stream";
			return 1;
		}
		reserve(gh[0], 1); // the reference reserves {size, hardware threads}; tests always pass threads=1
		count = gh[0];
		max_level = (int16_t)gh[3];
		entry = (uint32_t)gh[4];
		for (size_t i = 0; i != count; ++i) {
			size_t nb = node_bytes(lv[i]);
			lists[i].assign((nb - 10) / 4, 0);
			if (!get(&keys[i], 8) || !get(&levels[i], 2) || !get(lists[i].data(), nb - 10)) {
				err = "Failed to pull nodes from the stream";
				return 1;
			}
		}
		if (count != rows) {
			err = "Index size and the number of vectors doesn't match";
			return 1;
		}
		std::memcpy(vectors.data(), nvec.data(), rows * cols);
		// reindex_keys_ with enable_key_lookups=false: only the free ring is rebuilt
		size_t removed = 0;
		for (size_t i = 0; i != count; ++i)
			removed += keys[i] == FREE_KEY;
		tombstones = removed;
		if (removed) {
			free_keys.clear();
			free_keys.reserve(removed);
			for (size_t i = 0; i != count; ++i)
				if (keys[i] == FREE_KEY)
					free_keys.push((uint32_t)i);
		}
		return 0;
	}

	static orc_index orc_index_with(size_t dim, int metric, size_t M, size_t M0, size_t efc, size_t efs, int order,
	                                int wave) {
		orc_index x;
		x.dim = dim;
		x.metric = metric;
		x.M = M;
		x.M0 = M0;
		x.efc = efc;
		x.efs = efs;
		x.order = order;
		x.wave = wave;
		x.inv_log_m = 1.0 / std::log((double)M); // index.hpp:3549
		return x;
	}

	// ------------------------------------------------------------------ batch-synchronous bulk build
	// Restates the GPU engine's build (duckdb-vss_amd/csrc/vss_engine.hip): nodes are
	// inserted in batches against a frozen graph (phase A = the reference's search_to_insert_ + refine_ per
	// node and level), then all reverse links of the batch are applied per target list in ascending slot order
	// (phase B = the reference's reconnect step).  With every batch a singleton this is the reference's
	// sequential add() exactly.
	struct Request {
		uint32_t target;
		int level;
		uint32_t source;
		float d;
		size_t row; // index of the source row in the batch call (its vector is the `value` of the reverse link)
	};

	// solo_row: a row that re-links the current entry slot (its lists are blank while it is being re-linked, so batch
	// mates descending from the entry would find nothing but the entry itself): it runs alone, like a level promotion.
	static std::vector<size_t> schedule(size_t existing, int cur_max_level, const int16_t *lv, size_t n, size_t max_batch,
	                                    size_t growth_div, size_t solo_row = ~(size_t)0) {
		std::vector<size_t> sizes;
		size_t i = 0, cur = existing;
		int ml = cur_max_level;
		while (i < n) {
			size_t b = 1;
			if (cur != 0) {
				b = std::max<size_t>(1, std::min(max_batch, cur / growth_div));
				size_t take = 0;
				while (take < b && i + take < n) {
					if (lv[i + take] > ml || i + take == solo_row) {
						if (take == 0)
							take = 1;
						break;
					}
					take++;
				}
				b = take;
			}
			for (size_t j = 0; j != b; ++j)
				ml = std::max<int>(ml, lv[i + j]);
			sizes.push_back(b);
			i += b;
			cur += b;
		}
		return sizes;
	}

	// Rows whose slot comes from the free ring follow the reference's update() path (index_dense.hpp:1766-1793,
	// index.hpp:2801-2859): lists zeroed (level kept), reconnected from the global entry while the OLD vector is still
	// in place, key and vector replaced afterwards; no level is drawn for them.
	int build_batch(const int64_t *in_keys, const float *in_vecs, size_t n, size_t max_batch, size_t growth_div) {
		struct Row {
			size_t slot;
			bool reuse;
		};
		std::vector<Row> rows(n);
		std::vector<int16_t> lv(n);
		size_t first = count, n_new = 0;
		// the ring is popped row by row exactly as a sequence of add() calls would (index_dense.hpp:1767-1771)
		FreeRing ring_backup = free_keys;
		for (size_t i = 0; i != n; ++i) {
			if (slot_lookup.count(in_keys[i])) {
				free_keys = ring_backup;
				err = "Duplicate keys not allowed in high-level wrappers";
				return 1;
			}
			uint32_t free_slot = FREE_SLOT;
			free_keys.try_pop(free_slot);
			if (free_slot != FREE_SLOT)
				rows[i] = {free_slot, true};
			else
				rows[i] = {first + n_new++, false};
		}
		if (first + n_new > capacity) {
			free_keys = ring_backup;
			err = "Reserve capacity ahead of insertions!";
			return 1;
		}
		for (size_t i = 0; i != n; ++i) {
			const size_t slot = rows[i].slot;
			if (rows[i].reuse) {
				lv[i] = levels[slot];
			} else {
				lv[i] = rng.level(inv_log_m);
				node_make(slot, in_keys[i], lv[i]);
				slot_lookup.emplace(in_keys[i], (uint32_t)slot);
				std::memcpy(vectors.data() + slot * dim, in_vecs + i * dim, dim * sizeof(float));
			}
		}
		size_t solo_row = ~(size_t)0;
		for (size_t i = 0; i != n; ++i)
			if (rows[i].reuse && rows[i].slot == entry)
				solo_row = i;
		std::vector<size_t> sizes = schedule(first, max_level, lv.data(), n, max_batch, growth_div, solo_row);
		size_t done = 0;
		count = first + n_new;
		for (size_t b : sizes) {
			std::vector<Request> reqs;
			std::vector<std::pair<size_t, std::vector<uint32_t>>> parked;
			int16_t ml_before = max_level;
			size_t entry_before = entry;
			for (size_t j = 0; j != b; ++j) { // update(): zero the tapes of the reused nodes of this batch (:2837-2840)
				const Row &r = rows[done + j];
				if (!r.reuse)
					continue;
				std::fill(lists[r.slot].begin(), lists[r.slot].end(), 0u);
				if (keys[r.slot] == FREE_KEY)
					tombstones--;
				keys[r.slot] = 0;
			}
			for (size_t j = 0; j != b; ++j) {
				const Row &r = rows[done + j];
				const size_t slot = r.slot;
				const float *value = in_vecs + (done + j) * dim;
				if (slot == 0 && !r.reuse && first == 0 && done + j == 0) {
					entry = 0;
					max_level = levels[0];
					continue;
				}
				int16_t target = levels[slot];
				size_t closest = search_for_one(value, entry_before, ml_before, target);
				for (int level = std::min<int>(target, ml_before); level >= 0; --level) {
					if (wave)
						wave_level_search(value, closest, slot, level, add_top_limit(), true);
					else
						search_to_insert(value, closest, slot, level, add_top_limit());
					closest = connect_new_node(slot, level);
					const uint32_t *nb = list(slot, level);
					for (uint32_t i = 0; i != nb[0]; ++i)
						if (nb[1 + i] != slot)
							reqs.push_back({nb[1 + i], level, (uint32_t)slot, top.e[i].d, done + j});
				}
				if (!r.reuse && target > ml_before) { // only possible in a singleton batch
					entry = slot;
					max_level = target;
				}
				if (r.reuse) { // a reused node stays reachable through stale links: the rest of the batch must keep
					parked.emplace_back(slot, lists[slot]); // seeing its lists blank (phase A reads a frozen graph)
					std::fill(lists[slot].begin(), lists[slot].end(), 0u);
				}
			}
			for (auto &p : parked)
				lists[p.first].swap(p.second);
			std::stable_sort(reqs.begin(), reqs.end(), [](const Request &a, const Request &b) {
				if (a.level != b.level)
					return a.level < b.level;
				if (a.target != b.target)
					return a.target < b.target;
				return a.source < b.source;
			});
			for (const Request &r : reqs)
				reconnect_one(r.target, r.source, in_vecs + r.row * dim, r.level);
			for (size_t j = 0; j != b; ++j) { // update() epilogue: new key, then the vector (:2850, index_dense.hpp:1777-1781)
				const Row &r = rows[done + j];
				if (!r.reuse)
					continue;
				keys[r.slot] = in_keys[done + j];
				slot_lookup.emplace(in_keys[done + j], (uint32_t)r.slot);
				std::memcpy(vectors.data() + r.slot * dim, in_vecs + (done + j) * dim, dim * sizeof(float));
			}
			done += b;
		}
		return 0;
	}
};

// ---------------------------------------------------------------------------------------------
// C surface
// ---------------------------------------------------------------------------------------------
extern "C" {

orc_index *orc_create(uint64_t dim, int metric, uint64_t M, uint64_t M0, uint64_t efc, uint64_t efs) {
	return new orc_index(orc_index::orc_index_with(dim, metric, M, M0, efc, efs, 0, 0));
}
void orc_destroy(orc_index *h) {
	delete h;
}
const char *orc_last_error(orc_index *h) {
	return h->err.c_str();
}
int orc_reserve(orc_index *h, uint64_t members, uint64_t threads) {
	return h->reserve(members, threads) ? 0 : 1;
}
int orc_add(orc_index *h, int64_t key, const float *vec, uint64_t *stats) {
	return h->add(key, vec, stats);
}
uint64_t orc_search(orc_index *h, const float *q, uint64_t k, uint64_t ef, int exact, int64_t *keys, float *dists,
                    uint64_t *stats) {
	return h->search(q, k, ef, exact != 0, keys, dists, stats);
}
uint64_t orc_search_filtered(orc_index *h, const float *q, uint64_t k, uint64_t ef, const uint64_t *allowed,
                             uint64_t n_bits, int64_t *keys, float *dists, uint64_t *stats) {
	h->allowed = allowed;
	h->allowed_bits = n_bits;
	uint64_t n = h->search(q, k, ef, false, keys, dists, stats);
	h->allowed = nullptr;
	h->allowed_bits = 0;
	return n;
}
uint64_t orc_remove(orc_index *h, int64_t key) {
	return h->remove(key);
}
int orc_compact(orc_index *h) {
	h->compact();
	return 0;
}
int orc_compact_dropping(orc_index *h) {
	h->compact_dropping();
	return 0;
}
int orc_compact_reordering(orc_index *h) {
	h->compact_reordering();
	return 0;
}
uint64_t orc_size(orc_index *h) {
	return h->count - h->free_keys.size();
}
uint64_t orc_nodes(orc_index *h) {
	return h->count;
}
uint64_t orc_capacity(orc_index *h) {
	return h->capacity;
}
uint64_t orc_max_level(orc_index *h) {
	return h->count ? (uint64_t)h->max_level : 0;
}
void orc_level_stats(orc_index *h, uint64_t level, uint64_t *out4) { // index.hpp:3010-3027 (incl. quirk Q5)
	uint64_t nodes = 0, edges = 0, bytes = 0;
	uint64_t nbytes = level ? (4 + 4 * h->M) : (4 + 4 * h->M0);
	for (size_t i = 0; i != h->count; ++i) {
		if ((uint64_t)h->levels[i] < level)
			continue;
		nodes++;
		edges += h->list(i, (int)level)[0];
		bytes += 10 + nbytes;
	}
	out4[0] = nodes;
	out4[1] = edges;
	out4[2] = nodes * (level ? h->M0 : h->M);
	out4[3] = bytes;
}
uint64_t orc_serialized_length(orc_index *h) {
	return h->serialized_length();
}
int64_t orc_save(orc_index *h, uint8_t *buf, uint64_t cap) {
	return h->save(buf, cap);
}
int orc_load(orc_index *h, const uint8_t *buf, uint64_t len) {
	return h->load(buf, len);
}
float orc_distance(int metric, const float *a, const float *b, uint64_t dim) {
	return dist_reference_order(metric, a, b, dim);
}

// ---- oracle-only extensions (not exported by the reference shim) ----
// orc_compact_dropping / orc_compact_reordering (above): the engine's two compaction forms, see compact_dropping() and
// compact_reordering()

/* order: 0 reference / 1 wave summation order; wave: 0 reference / 1 kernel candidate lists */
// model of the engine's register queue: cap = entries (0 = off); returns and clears "the last searches overflowed it"
void orc_set_register_queue(orc_index *h, uint64_t cap) {
	h->regq_cap = cap;
	h->regq_overflow = false;
	h->regq_drops = 0;
}
uint64_t orc_register_queue_state(orc_index *h, uint64_t *drops) {
	const uint64_t overflowed = h->regq_overflow ? 1 : 0;
	if (drops)
		*drops = h->regq_drops;
	h->regq_overflow = false;
	h->regq_drops = 0;
	return overflowed;
}
/* the pipelined level search's successor rule, checked against the plain order (see orc_index::pipe_check) */
void orc_set_pipeline_check(orc_index *h, int on) {
	h->pipe_check = on != 0;
	h->pipe_stats[0] = h->pipe_stats[1] = h->pipe_stats[2] = 0;
}
void orc_pipeline_check_state(orc_index *h, uint64_t *out3) {
	out3[0] = h->pipe_stats[0], out3[1] = h->pipe_stats[1], out3[2] = h->pipe_stats[2];
}
void orc_set_mode(orc_index *h, int order, int wave) {
	h->order = order;
	h->wave = wave;
}
float orc_distance_wave(int metric, const float *a, const float *b, uint64_t dim) {
	return dist_wave_order(metric, a, b, dim);
}
/* array_distance (fn 0) / array_cosine_distance (1) / array_negative_inner_product (2) over rows x dim floats, one
 * sequential f32 accumulation per row — SURVEY Appendix B's statement of DuckDB core's functions (named at reference
 * hnsw_index.cpp:659-673; their source is NOT in the reference tree: PARITY UNPINNED).  b = rows x dim, or one vector when
 * b_const.  bench.py --config a13 times this loop on one host thread beside vss_distance_batch on the same chunks. */
void orc_array_function(int fn, const float *a, const float *b, int b_const, uint64_t rows, uint64_t dim, float *out) {
	for (uint64_t r = 0; r != rows; ++r) {
		const float *x = a + r * dim, *y = b_const ? b : b + r * dim;
		float ab = 0.f, a2 = 0.f, b2 = 0.f;
		if (fn == 0) {
			for (uint64_t i = 0; i != dim; ++i) {
				const float t = x[i] - y[i];
				ab += t * t;
			}
			out[r] = std::sqrt(ab);
		} else if (fn == 2) {
			for (uint64_t i = 0; i != dim; ++i)
				ab += x[i] * y[i];
			out[r] = -ab;
		} else {
			for (uint64_t i = 0; i != dim; ++i) {
				ab += x[i] * y[i];
				a2 += x[i] * x[i];
				b2 += y[i] * y[i];
			}
			float sim = ab / std::sqrt(a2 * b2);
			sim = sim > 1.f ? 1.f : (sim < -1.f ? -1.f : sim);
			out[r] = 1.f - sim;
		}
	}
}
/* first n draws of the level generator for connectivity M (fresh default-seeded engine) */
void orc_draw_levels(uint64_t M, uint64_t n, int16_t *out) {
	LevelRng r;
	double inv = 1.0 / std::log((double)M);
	for (uint64_t i = 0; i != n; ++i)
		out[i] = r.level(inv);
}
/* batch sizes the bulk build uses; returns number of batches (out may be NULL to count) */
uint64_t orc_schedule(uint64_t existing, int cur_max_level, const int16_t *levels, uint64_t n, uint64_t max_batch,
                      uint64_t growth_div, uint64_t *out) {
	auto s = orc_index::schedule(existing, cur_max_level, levels, n, max_batch, growth_div);
	if (out)
		for (size_t i = 0; i != s.size(); ++i)
			out[i] = s[i];
	return s.size();
}
int orc_build_batch(orc_index *h, const int64_t *keys, const float *vecs, uint64_t n, uint64_t max_batch,
                    uint64_t growth_div) {
	return h->build_batch(keys, vecs, n, max_batch, growth_div);
}
/* raw graph access for structural tests */
int orc_node_level(orc_index *h, uint64_t slot) {
	return h->levels[slot];
}
int64_t orc_node_key(orc_index *h, uint64_t slot) {
	return h->keys[slot];
}
uint64_t orc_neighbors(orc_index *h, uint64_t slot, int level, uint32_t *out) {
	const uint32_t *nb = h->list(slot, level);
	for (uint32_t i = 0; i != nb[0]; ++i)
		out[i] = nb[1 + i];
	return nb[0];
}
uint64_t orc_entry_slot(orc_index *h) {
	return h->entry;
}
void orc_counters(orc_index *h, uint64_t *out2) {
	out2[0] = h->computed;
	out2[1] = h->cycles;
}
}
