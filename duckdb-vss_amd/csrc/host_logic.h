// host_logic.h — the pure host-side logic of the engine: no HIP, no device state, so that it can be compiled and tested
// on a CPU (tests/host_logic_probe.cpp, tests/test_host_logic.py) against the oracle and the reference build.
//
//   LevelRng        the reference's level generator (libstdc++ minstd_rand0 + generate_canonical<double,53>)
//   FreeRing        usearch's ring_gt as index_dense uses it for freed slots, wrap quirk included
//   KeyMap          rowid -> slot (lazy; deletes and duplicate checks)
//   batch_schedule  how a bulk build is cut into batch-synchronous steps
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "visited_compact.h"

namespace vss {
namespace host {

constexpr int64_t FREE_KEY_HOST = 0x7FFFFFFFFFFFFFFFll; // = VSS_FREE_KEY (include/vssgpu.h), usearch's free_key_
constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;               // = vss::EMPTY_SLOT

inline size_t ceil_pow2(size_t v) {
	size_t p = 1;
	while (p < v)
		p <<= 1;
	return p;
}
inline uint32_t log2u(size_t v) {
	uint32_t l = 0;
	while ((size_t(1) << l) < v)
		l++;
	return l;
}

// Level generator: libstdc++'s std::default_random_engine + uniform_real_distribution<double>, as used by
// usearch choose_random_level_ (index.hpp:3723-3727).  One stream per index (the reference keeps one per thread
// context, all identically seeded — SURVEY A.2); restarted by every growing reserve.
struct LevelRng {
	uint64_t x = 1;
	uint32_t next() {
		x = (x * 16807ull) % 2147483647ull;
		return (uint32_t)x;
	}
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
	int level(double inv_log_m) {
		double r = -std::log(canonical()) * inv_log_m;
		return (int)(int16_t)r;
	}
	// the level as the engine stores it (u8)
	uint8_t stored_level(double inv_log_m) {
		const int lv = level(inv_log_m);
		return (uint8_t)(lv < 0 ? 0 : lv > 255 ? 255 : lv);
	}
};
inline double inverse_log_connectivity(uint64_t M) { // usearch index.hpp:3549
	return 1.0 / std::log((double)M);
}

// rowid -> slot, open addressing; built lazily (only deletes and duplicate checks need it)
struct KeyMap {
	std::vector<int64_t> k;
	std::vector<uint32_t> v;
	size_t mask = 0, used = 0;
	bool ready = false;
	static uint64_t hash(int64_t key) {
		uint64_t z = (uint64_t)key + 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		return z ^ (z >> 31);
	}
	void init(size_t n) {
		size_t cap = ceil_pow2(std::max<size_t>(16, n * 2));
		k.assign(cap, FREE_KEY_HOST);
		v.assign(cap, 0);
		mask = cap - 1;
		used = 0;
		ready = true;
	}
	void grow() {
		std::vector<int64_t> ok;
		std::vector<uint32_t> ov;
		ok.swap(k);
		ov.swap(v);
		init(ok.size());
		for (size_t i = 0; i != ok.size(); ++i)
			if (ok[i] != FREE_KEY_HOST && ov[i] != NO_SLOT)
				put(ok[i], ov[i]);
	}
	void put(int64_t key, uint32_t slot) {
		if ((used + 1) * 2 > k.size())
			grow();
		size_t h = hash(key) & mask;
		while (k[h] != FREE_KEY_HOST && k[h] != key)
			h = (h + 1) & mask;
		if (k[h] == FREE_KEY_HOST)
			used++;
		k[h] = key;
		v[h] = slot;
	}
	bool find(int64_t key, uint32_t &slot) const {
		size_t h = hash(key) & mask;
		while (k[h] != FREE_KEY_HOST) {
			if (k[h] == key) {
				slot = v[h];
				return slot != NO_SLOT;
			}
			h = (h + 1) & mask;
		}
		return false;
	}
	void erase(int64_t key) { // keep the key as a probe-chain marker, drop the slot
		size_t h = hash(key) & mask;
		while (k[h] != FREE_KEY_HOST) {
			if (k[h] == key) {
				v[h] = NO_SLOT;
				return;
			}
			h = (h + 1) & mask;
		}
	}
};

// The free list of tombstoned slots.  Restates usearch's ring_gt (index.hpp:1150-1277) as index_dense uses it
// (free_keys_, index_dense.hpp:463) INCLUDING its size() == 0 when the ring is exactly full: the order in which removed
// slots are handed back to later inserts is part of the reference's observable behaviour (which slot a row lands in).
struct FreeRing {
	std::vector<uint32_t> el;
	size_t cap = 0, head = 0, tail = 0;
	bool empty = true;
	size_t size() const {
		if (empty)
			return 0;
		return head >= tail ? head - tail : cap - (tail - head);
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
		std::vector<uint32_t> grown(n);
		size_t i = 0;
		while (try_pop(grown[i]))
			i++;
		el.swap(grown);
		cap = n, head = i, tail = 0;
		empty = i == 0;
		return true;
	}
	void clear() {
		head = tail = 0;
		empty = true;
	}
};


// Batch schedule of the bulk build (mirrored by oracle/hnsw_oracle.cpp `schedule`): batch = clamp(nodes / growth_div, 1,
// max_batch); a row whose level exceeds the current top level runs alone and becomes the entry (index.hpp:2769-2772).
// solo_row: the row that re-links the current entry slot, if any.  Its lists are blank while it is being re-linked, so
// batch mates descending from the entry would find nothing but the entry: it runs alone, like a level promotion.
inline std::vector<uint64_t> batch_schedule(uint64_t existing, int cur_max_level, const uint8_t *lv, uint64_t n,
                                      uint64_t max_batch, uint64_t growth_div, uint64_t solo_row = ~0ull) {
	std::vector<uint64_t> sizes;
	uint64_t i = 0, cur = existing;
	int ml = cur_max_level;
	while (i < n) {
		uint64_t b = 1;
		if (cur != 0) {
			b = std::max<uint64_t>(1, std::min(max_batch, cur / growth_div));
			uint64_t take = 0;
			while (take < b && i + take < n) {
				if ((int)lv[i + take] > ml || i + take == solo_row) {
					if (take == 0)
						take = 1;
					break;
				}
				take++;
			}
			b = take;
		}
		for (uint64_t j = 0; j != b; ++j)
			ml = std::max<int>(ml, lv[i + j]);
		sizes.push_back(b);
		i += b;
		cur += b;
	}
	return sizes;
}

// ---------------------------------------------------------------------------------------------------------
// Capacity of a walker's visited set (log2 of its cells; no reference counterpart — usearch's growing_hash_set_gt grows on
// demand, index.hpp:1018-1144; here a query that fills its table beyond 7/8 is re-run with a larger one).
//   per_limit  cells per entry of the search / insert limit: 64 is a table no ordinary query fills (the build, searches up to
//              limit 128 and beyond 256); searches with limits 129-256 take 32 — half the table, which stays in LDS (2^13
//              cells = 32 KiB, four walkers per workgroup) where 64 would push it to HBM and make every probe round an L2 /
//              memory round trip (the 2-74 queries of 10 240 that outgrow it are re-run)
//   bump       added after an overflow (the retry also carries a floor: strictly larger than the table that overflowed)
//   max_log2   "every node fits below 7/8": a table of that size cannot overflow
// ---------------------------------------------------------------------------------------------------------
inline uint64_t search_cells_per_limit(uint64_t limit) {
	return (limit > 128 && limit <= 256) ? 32 : 64;
}
inline uint32_t visited_set_log2(uint64_t limit, uint32_t bump, uint64_t M0, uint64_t list_cap_max, uint64_t per_limit,
                                 uint32_t max_log2) {
	uint64_t cap = ceil_pow2(per_limit * std::max<uint64_t>(std::min<uint64_t>(limit, 1u << 20), 2 * M0));
	cap = std::max<uint64_t>(cap, ceil_pow2(8ull * list_cap_max));
	cap = std::max<uint64_t>(cap, 1024);
	return std::min<uint32_t>(log2u(cap) + bump, max_log2);
}

// The compact visited set (visited_compact.h; 16-bit cells: tag + displacement, exact): taken by a launch of the workgroup
// engine whose 32-bit table would not fit LDS, when the limit is one of the 8-register list's (257-512: the only instantiation
// that carries the code), every slot fits 25 bits (24 until round 6) and this is the query's FIRST pass at the host's hands (a
// query whose set outgrows the cells — too many visits, a displacement beyond its bits — moves it to a plain table in HBM on the
// device, VisitedSet::migrate; one that outgrows that too is re-run by the host with the plain 32-bit table like any other
// overflow).  Returns the set's FORM (visited_compact.h: log2 of its cells — twice the words of the 32-bit table of
// 2^lds_table_log2 words it lies over — in bits 0-7, the key bits in bits 8-15 when they are not 24), or 0 = the plain set.
inline uint32_t compact_visited_cells_log2(bool plain_table_fits_lds, bool solo_shape, bool register_list, uint64_t limit,
                                           uint64_t nodes, bool first_pass, uint32_t lds_table_log2) {
	if (plain_table_fits_lds || solo_shape || !register_list || limit <= 256 || !first_pass)
		return 0;
	if (nodes > (1ull << compact_visited::KEY_BITS_MAX))
		return 0;
	const uint32_t cells_log2 = lds_table_log2 + 1;
	const uint32_t form = nodes > (1ull << compact_visited::KEY_BITS) ? compact_visited::make_form(cells_log2, compact_visited::KEY_BITS_MAX)
	                                                                  : cells_log2;
	return compact_visited::form_ok(form) ? form : 0;
}

// ---------------------------------------------------------------------------------------------------------
// Which shape of the search engine answers a launch of n queries (no reference counterpart —
// results never depend on it):
//   workgroups  k_search: persistent 1024-thread workgroups, walkers + scoring waves exchanging rows through LDS mailboxes
//   solo        k_search_solo<.., 1>: one self-scoring wave per query
//   team        k_search_solo<.., 8>: the walking wave + helper waves behind two workgroup barriers per expansion
//   crew        (round 4) not a kernel of its own: inside k_search the LAST walker of a workgroup runs the scoring waves
//               behind two workgroup barriers instead of the mailboxes — from the first expansion on when a launch has
//               at most one query per compute unit (wide rows: the one-query probe and the join chunks at 768 dims),
//               and in the drain of every larger launch
// and what the latency-bound launches touch ahead (RowTouch: lines per row in bits 0-7; ListTouch: TOUCH_LISTS_BIT).
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t TOUCH_LISTS_BIT = 0x100u;    // = vss::TOUCH_LISTS (hnsw_kernels.h)
constexpr uint32_t LDS_BYTES_PER_CU = 160u * 1024;
struct SearchShapePolicy {
	uint32_t solo_mode = 1;           // 0 never, 1 automatic, 2 always (vss_set_search_solo)
	uint32_t solo_max_queries = 32;   // automatic: the one-wave shape up to this many queries per launch ...
	uint32_t solo_max_bytes = 32 * 1024; // ... over rows narrow enough that a level-0 list of them is within this
	bool team = true;                 // vss_set_search_team
	uint32_t n_cus = 256;             // a team wants a compute unit per query
	bool touch_rows = true, touch_lists = true;
	uint32_t touch_max_queries = 256; // touches cost bandwidth: only launches that cannot be bound by it
	bool force_looping = false;
	uint32_t team_box_bytes = 528;    // = vss::TEAM_BOX_BYTES
	bool crew = true;                 // vss_set_search_crew
	uint32_t engine_walkers = 0;      // vss_set_search_params: walkers per workgroup forced (0 = from the launch size)
};
struct SearchShape {
	bool solo = false, team = false;
	uint32_t touch_lines = 0;
	bool crew = false;  // workgroup engine: the last walker of a workgroup may run its scoring waves as a crew
	bool roomy = false; // one walker per compute unit from the start: the visited set may take up to 64 KiB of LDS
};
// chunks per lane of the unrolled kernel variants (0 = the looping kernels): V float4 chunks per row over G lanes
inline uint32_t chunks_per_lane(uint64_t V, uint64_t G, bool force_looping) {
	return (V % G == 0 && !force_looping) ? (uint32_t)(V / G) : 0u;
}
// team variants exist for one chunk per lane and for the looping kernels (wide rows stay with the workgroup engine: a team
// of 8 measured slower than its 15 scoring waves, profiles/r03s_engine_shapes_by_batch_3m768_wide_rows.txt)
inline bool team_variant_exists(uint32_t nch) {
	return nch <= 1;
}
// rows of at most this many 128-byte lines are touched ahead by a team's helpers (= vss::team_touch_max_lines)
inline uint32_t team_touch_lines_max(uint32_t nch) {
	return nch > 0 ? 8 * nch : 8;
}
inline bool wants_solo(const SearchShapePolicy &p, uint32_t n, uint64_t M0, uint64_t V, uint64_t G) {
	if (p.solo_mode != 1)
		return p.solo_mode == 2;
	if (M0 * V * 16 > p.solo_max_bytes)
		return false;
	return n <= p.solo_max_queries ||
	       (p.team && team_variant_exists(chunks_per_lane(V, G, p.force_looping)) && n <= p.n_cus);
}
// the workgroup engine with ONE walker per workgroup from the start (a launch of at most one query per compute unit, or forced)
inline bool one_walker_launch(const SearchShapePolicy &p, uint32_t n) {
	return p.engine_walkers ? p.engine_walkers == 1 : n <= p.n_cus;
}
// one walker per compute unit — the solo shape, or a one-walker launch of the workgroup engine — has the unit's LDS to itself:
// the visited set may take up to 64 KiB (the rule the engine's launch path and choose_search_shape share)
inline bool roomy_visited_set(const SearchShapePolicy &p, bool solo, uint32_t n) {
	return solo || one_walker_launch(p, n);
}
// solo_lds_bytes: the dynamic LDS of one solo workgroup for this launch (visited set, staged query, id / distance buffers)
inline SearchShape choose_search_shape(const SearchShapePolicy &p, uint32_t n, uint64_t M0, uint64_t V, uint64_t G,
                                       uint32_t solo_lds_bytes) {
	SearchShape s;
	s.solo = wants_solo(p, n, M0, V, G);
	if (!s.solo) {
		s.crew = p.crew;
		// a launch of at most one query per compute unit runs one walker per workgroup: alone from the first expansion on,
		// i.e. a latency chain — ListTouch as in the solo shape (a 128-byte line per accepted row; RowTouch never: the rows
		// the engine serves are wide, and pulling 64 of them ahead would swamp the compute unit's fill rate)
		const bool one_walker = one_walker_launch(p, n);
		s.roomy = roomy_visited_set(p, false, n);
		if (s.crew && one_walker && n <= p.n_cus && n <= p.touch_max_queries && p.touch_lists)
			s.touch_lines = TOUCH_LISTS_BIT;
		return s;
	}
	s.roomy = true;
	const uint32_t nch = chunks_per_lane(V, G, p.force_looping);
	s.team = p.team && team_variant_exists(nch) && n <= p.n_cus && solo_lds_bytes + p.team_box_bytes <= LDS_BYTES_PER_CU;
	if (n <= p.touch_max_queries) {
		const uint32_t row_lines = (uint32_t)((V * 16 + 127) / 128);
		const uint32_t max_lines = s.team ? team_touch_lines_max(nch) : 4u; // (the lone wave: RowTouch<.., 4>)
		if (p.touch_rows && row_lines <= max_lines)
			s.touch_lines = row_lines;
		if (p.touch_lists)
			s.touch_lines |= TOUCH_LISTS_BIT;
	}
	return s;
}

} // namespace host
} // namespace vss
