// hnsw_kernels.h — the HNSW traversal, insertion and link-repair kernels (gfx950, one wavefront per work item).
//
// Reference behaviour being reproduced (usearch 2.12.0 as vendored by duckdb-vss, paths under
// /root/reference/src/include/usearch):
//   descend()            = index_gt::search_for_one_            index.hpp:3809-3847
//   level_search()       = search_to_insert_ / search_to_find_in_base_   index.hpp:3855-3921 / 3929-3998
//   refine_candidates()  = refine_                               index.hpp:4027-4063
//   k_search             = index_gt::search                      index.hpp:2876-2930
//   k_build_phase_a      = connect_node_across_levels_ minus the reverse links   index.hpp:3635-3675
//                          (and index_gt::update's re-link of a reused slot, index.hpp:2801-2859)
//   k_build_phase_b      = reconnect_neighbor_nodes_ (the reverse links)          index.hpp:3678-3721
// The CPU mirror of exactly these kernels is oracle/hnsw_oracle.cpp with order=1, wave=1.
// Latency helpers that change no decision: search teams (W scoring waves per query) and ListPrefetch, below.
//
// Graph layout in HBM (struct of arrays, fixed stride, empty cells = 0xFFFFFFFF, lists packed at the front):
//   vectors   [capacity][V] float4         row-major FLOAT[dim] payload, zero padded to 16 bytes
//   links0    [capacity][M0] u32           level-0 neighbour lists (M0 = 32 -> one 128-byte line per node)
//   links_up  [n_upper_lists][M] u32       lists of levels >= 1; node `s` owns lists upper_off[s] .. +level(s)-1
//   keys      [capacity] i64               DuckDB row ids; VSS_FREE_KEY marks a tombstone
#pragma once
#include "wave_primitives.h"
#include <type_traits>

namespace vss {

constexpr int64_t FREE_KEY = 0x7FFFFFFFFFFFFFFFll;

struct GraphView {
	RowSpace sp;
	uint32_t dim;
	uint32_t M, M0;
	uint32_t *links0;
	uint32_t *links_up;
	const uint32_t *upper_off;
	const int64_t *keys;
	uint32_t list_id_base; // list id of upper list u is list_id_base + u (level-0 list of slot s has id s)
	// optional result predicate (usearch filtered_search, index_dense.hpp:625-629): bitmap over row ids, bit set = admitted
	const unsigned long long *filter;
	uint64_t filter_bits;
	// a list of this graph may name one slot twice (slots have been re-used, see mark_first_visit)
	uint32_t twins;

	// `key != free_key [&& predicate(key)]` — index_dense.hpp:1817-1824
	__device__ __forceinline__ bool admitted(uint32_t slot) const {
		const int64_t key = keys[slot];
		if (key == 0x7FFFFFFFFFFFFFFFll)
			return false;
		if (!filter)
			return true;
		return key >= 0 && (uint64_t)key < filter_bits && ((filter[key >> 6] >> (key & 63)) & 1ull);
	}

	__device__ __forceinline__ uint32_t *list_ptr(uint32_t slot, int level) const {
		return level == 0 ? links0 + (size_t)slot * M0 : links_up + ((size_t)upper_off[slot] + (level - 1)) * M;
	}
	__device__ __forceinline__ uint32_t list_cap(int level) const {
		return level == 0 ? M0 : M;
	}
};

// LDS carve-up of one wave (all offsets multiples of 16 bytes)
struct WaveLds {
	VisitedSet visited;
	float4 *q;      // the staged query / the node being inserted
	float4 *q2;     // a second staged row (refine_)
	uint32_t *ids;  // neighbour ids being scored        [list_cap_max rounded up to 64]
	float *dist;    // their distances                    [same]
	float *cand_d;  // candidate list dumped from registers [cand_cap]
	uint32_t *cand_s;
	uint32_t *kept_s; // refine_ output                   [list_cap_max + 1]
	float *kept_d;
	uint32_t touch_lines = 0; // solo search kernel: bits 0-7 = 128-byte lines per row to pull into L2 ahead of time (RowTouch; 0 = off), TOUCH_LISTS = ListTouch
	// Workgroup engine, limits of 257-512 (the compact visited set): four words in LDS — {address of this walker's table in HBM
	// (64 bits), log2 of its cells, "this query's set has moved there"} — or nullptr.  In LDS, not in this struct: the walker's
	// kernel sits at its scalar-register limit, and the words are read on the rare path only (gather_neighbors).
	uint32_t *spill_box = nullptr;
};

struct WorkCounters {
	uint32_t distances;
	uint32_t cycles;
#ifdef VSS_PHASE_TIMERS // debug builds only (tools/gpu_profile.py): shader-clock ticks per phase of level_search
	// (32-bit: twelve 64-bit accumulators do not fit the scalar registers next to the engine's walker and end up in scratch)
	uint32_t t_pick, t_gather, t_dist, t_accept, t_descend, t_total;
	uint32_t t_sync1, t_look, t_slice, t_sync2, t_team_passes, t_solo_passes;
#endif
};

#ifdef VSS_PHASE_TIMERS
#define VSS_TICK(var) const unsigned long long var = __builtin_readcyclecounter()
#define VSS_ACC(field, a, b) wc.field += (uint32_t)((b) - (a))
#define VSS_COUNT(field, n) wc.field += (n)
#define VSS_WC_ARG , WorkCounters &wc
#define VSS_WC_PASS , wc
#define VSS_PHASE_STRIDE 12
#else
#define VSS_WC_ARG
#define VSS_WC_PASS
#define VSS_TICK(var)
#define VSS_ACC(field, a, b)
#define VSS_COUNT(field, n)
#endif

// ---------------------------------------------------------------------------------------------------------
// visits.set() for one chunk of a neighbour list, one id per lane: true on the lanes whose id was not visited before.
// A list can name the same slot twice (a stale link to a re-used slot plus its new reverse link:
// reconnect_neighbor_nodes_ appends without looking).  The reference's sequential visits.set() keeps the FIRST
// occurrence, and with tied distances that position decides the order in the candidate list.  Which of two lanes wins a
// compare-and-swap is the hardware's business (the LDS happens to serve the lowest lane first, the L2 does not), so a lane
// that lost against a twin in this very chunk hands the win to the lowest lane of the group.  Twins exist only in graphs
// whose slots have been re-used (the host knows: GraphView::twins); all others take the plain path.
// `bad` (compact form of the set only): non-zero on a lane whose key could not be placed — the caller reports an overflow.
__device__ __forceinline__ bool mark_first_visit(VisitedSet &visited, uint32_t id, bool have, bool twins, uint32_t &bad) {
	if (!twins) // every id of a list is distinct (no slot was ever re-used): a plain compare-and-swap decides
		return have && !visited.test_and_set(id, bad);
	const int seen = have ? visited.probe(id, bad) : VisitedSet::SEEN_BEFORE;
	bool fresh = seen == VisitedSet::INSERTED;
	unsigned long long lost = __ballot(seen == VisitedSet::LOST_TO_TWIN);
	while (lost) { // rare
		const uint32_t key = read_lane(id, __builtin_ctzll(lost));
		const bool mine = seen != VisitedSet::SEEN_BEFORE && id == key;
		const unsigned long long group = __ballot(mine);
		// (a stale read of a cell filled earlier also ends up here: then nobody of the group inserted the key)
		const bool inserted_now = __ballot(mine && seen == VisitedSet::INSERTED) != 0;
		if (mine)
			fresh = inserted_now && lane_id() == __builtin_ctzll(group);
		lost &= ~group;
	}
	return fresh;
}

// ---------------------------------------------------------------------------------------------------------
// Read one neighbour list and (optionally) filter it through the visited set; the surviving ids are packed,
// order preserved, into lds.ids.  Returns their number (wave-uniform) or -1 on visited-set overflow.
// `have_first`: the first 64 cells of the list were already fetched into `first` (one per lane, see ListPrefetch).
template <bool FILTER>
__device__ __forceinline__ int gather_neighbors(const GraphView &gv, WaveLds &lds, uint32_t slot, int level,
                                                bool have_first = false, uint32_t first = EMPTY_SLOT) {
	const int lane = lane_id();
	const uint32_t cap = gv.list_cap(level);
	const uint32_t *lp = gv.list_ptr(slot, level);
	int n = 0;
	uint32_t bad = 0;
	for (uint32_t off = 0; off < cap; off += 64) {
		uint32_t id;
		if (off == 0 && have_first)
			id = first;
		else
			id = (off + lane < cap) ? lp[off + lane] : EMPTY_SLOT;
		bool take = id != EMPTY_SLOT;
		if (FILTER) {
			take = mark_first_visit(lds.visited, id, take, gv.twins != 0, bad);
			if (lds.visited.compact && lds.spill_box && __ballot(bad != 0)) {
				// a key's displacement did not fit the compact set's bits: the set moves to this walker's table in HBM, and the
				// keys that did not fit go through the new table (first occurrence wins there as everywhere)
				lds.visited.migrate(reinterpret_cast<uint32_t *>((uintptr_t)lds.spill_box[0] | ((uintptr_t)lds.spill_box[1] << 32)),
				                    lds.spill_box[2]);
				if (lane == 0)
					lds.spill_box[3] = 1u;
				const bool again = bad != 0;
				uint32_t none = 0;
				const bool fresh = mark_first_visit(lds.visited, id, again, gv.twins != 0, none);
				take = again ? fresh : take;
				bad = 0;
			}
		}
		unsigned long long m = __ballot(take);
		if (take)
			lds.ids[n + __popcll(m & lanes_below(lane))] = id;
		n += __popcll(m);
	}
	if (FILTER) {
		lds.visited.count += n;
		if (lds.visited.count > lds.visited.limit) {
			if (!(lds.visited.compact && lds.spill_box))
				return -1;
			// the compact set is three quarters full: it moves to this walker's table in HBM and the search goes on
			lds.visited.migrate(reinterpret_cast<uint32_t *>((uintptr_t)lds.spill_box[0] | ((uintptr_t)lds.spill_box[1] << 32)),
			                    lds.spill_box[2]);
			if (lane == 0)
				lds.spill_box[3] = 1u;
			if (lds.visited.count > lds.visited.limit)
				return -1;
		}
		if (lds.visited.compact && __ballot(bad != 0)) // (compact form without a table to move to: a displacement did not fit its bits)
			return -1;
	}
	lds_sync(); // the ids are in LDS (global loads issued ahead of time — list requests, touches — stay in flight; the atomics on
	            // a visited set that lives in HBM have returned: their results were used above)
	return n;
}

// ---------------------------------------------------------------------------------------------------------
// An expansion costs two dependent HBM round trips: the neighbour list, then the rows it names.  The list of the
// candidate that will most likely be expanded NEXT (the best unexpanded entry once the current one is marked) is
// therefore requested while the current rows are still in flight, one cell per lane, and kept until it is used or a
// better guess replaces it.  Pure latency hiding: which lists are expanded, and in what order, does not change.
// A request must not WAIT for its load: the loaded dword is written to its register and nothing else — no select on it
// (`lane < cap ? load : EMPTY`, or `slot == i ? arrived : cells[i]`, makes hipcc wait for the load right there: round 3's
// look-ahead cost the walker two full memory round trips per expansion, 2.5-3k cycles, found with the phase timers of
// round 4).  So the address is clamped instead of the value, and the cells beyond the list's capacity are masked when the
// list is USED (find / cells_of).
struct ListPrefetch {
	uint32_t slot = EMPTY_SLOT; // whose list `cells` holds
	uint32_t cells = EMPTY_SLOT;
	uint32_t cap = 0;   // cells of that list (lanes beyond it hold a copy of cell 0)
	uint32_t fresh = 0; // bit 0: the rows this list names have not been touched yet (RowTouch)
	__device__ __forceinline__ void request(const GraphView &gv, uint32_t want, int level) {
		if (want == slot)
			return;
		cap = gv.list_cap(level);
		const uint32_t *lp = gv.list_ptr(want, level);
		cells = lp[(uint32_t)lane_id() < cap ? lane_id() : 0];
		slot = want;
		fresh = 1;
	}
	__device__ __forceinline__ uint32_t masked() const {
		return (uint32_t)lane_id() < cap ? cells : EMPTY_SLOT;
	}
};

// Round 3: K lists in flight instead of one.  The single guess above is right only when none of the rows being scored
// beats the runner-up; with the best TWO unexpanded entries requested while the rows are in flight, the runner-up's list
// is there as well when the best one turns out to be the row scored last.  A list costs one register per lane and one
// 128/256-byte load; round-robin replacement.  Still pure latency hiding.  (Measured and dropped: requesting again right
// after the accept phase — the two list scans cost the accept phase 500-700 cycles per expansion and the gather phase
// gained nothing: it is dominated by the visited-set probes, not by list latency.)
template <int K>
struct ListCache {
	static constexpr int slots = K;
	uint32_t slot[K];  // wave-uniform
	uint32_t cells[K]; // one cell per lane (lists of at most 64 cells; lanes beyond the capacity hold a copy of cell 0)
	uint32_t cap = 0;  // cells of the lists of this level
	uint32_t next = 0;
	uint32_t fresh = 0; // bit i: the rows list i names have not been touched yet (RowTouch)
	__device__ __forceinline__ ListCache() {
#pragma unroll
		for (int i = 0; i < K; ++i)
			slot[i] = EMPTY_SLOT, cells[i] = EMPTY_SLOT;
	}
	__device__ __forceinline__ bool find(uint32_t want, uint32_t &out) const {
		bool hit = false;
		out = EMPTY_SLOT;
#pragma unroll
		for (int i = 0; i < K; ++i)
			if (slot[i] == want) {
				out = cells[i];
				hit = true;
			}
		if ((uint32_t)lane_id() >= cap)
			out = EMPTY_SLOT;
		return hit;
	}
	__device__ __forceinline__ void request(const GraphView &gv, uint32_t want, int level) {
		bool have = false;
#pragma unroll
		for (int i = 0; i < K; ++i)
			have = have || slot[i] == want;
		if (have)
			return;
		cap = gv.list_cap(level);
		const uint32_t *lp = gv.list_ptr(want, level) + ((uint32_t)lane_id() < cap ? lane_id() : 0);
#pragma unroll
		for (int i = 0; i < K; ++i) // (no run-time register index: that would live in scratch memory; the branch is wave-uniform
			if (next == (uint32_t)i) { //  and each arm loads straight into its own register: nothing waits for the load here —
				const uint32_t *lpi = lp; // the pointer is made opaque per arm, or the identical loads are hoisted out of the
				asm volatile("" : "+v"(lpi)); // arms and selected into place, which waits)
				// (the opaque pointer has lost its address space: named again, or the load is a FLAT one — which counts on lgkmcnt
				//  as well, so that the next wait for an LDS read waits for this HBM round trip too)
				cells[i] = *(const __attribute__((address_space(1))) uint32_t *)(uintptr_t)lpi;
				slot[i] = want;
			}
		fresh |= 1u << next;
		next = next + 1 == (uint32_t)K ? 0u : next + 1;
	}
	__device__ __forceinline__ uint32_t cells_of(int i) const {
		return (uint32_t)lane_id() < cap ? cells[i] : EMPTY_SLOT;
	}
	__device__ __forceinline__ uint32_t fresh_bits() const {
		return fresh;
	}
	__device__ __forceinline__ void clear_fresh() {
		fresh = 0;
	}
};

// ---------------------------------------------------------------------------------------------------------
// Scorers: how the walking wave turns the ids gathered in lds.ids[0..n) into distances in lds.dist[0..n).
//   SoloScorer  the wave scores the rows itself (build kernels: one wave per inserted node / repaired list).
//   PoolScorer  the search engine (k_search): the rows are offered to the scoring waves of the workgroup.
// A row is reduced by the same lanes in the same order either way, so distances — and everything decided from them —
// keep their bits.  `before_loads` runs on the walking wave once the job is on offer and before its own row loads are
// issued (the place for loads that should overlap them, e.g. ListPrefetch).
// LATE = true (the solo search kernel): `before_loads` runs right AFTER the first pass's row loads have been issued instead —
// the wave would only wait for them meanwhile — so the look-ahead costs nothing on the critical path; its own list load
// queues behind the rows and has long arrived when the next expansion asks for it.
template <int MT, int NCH, int R, bool LATE = false>
struct SoloScorer {
	// the one-wave search kernel touches ahead itself: the rows of the cached lists (RowTouch) and the lists of the rows it
	// accepts (ListTouch)
	static constexpr bool touches_rows = LATE, touches_lists = LATE, helpers_touch = false;
	__device__ __forceinline__ void at_level(int) const { // (descend<.., AHEAD>: only a crew's scoring waves care)
	}
	__device__ __forceinline__ bool latency_mode() const { // (ListTouch is the host's decision alone: WaveLds::touch_lines)
		return false;
	}
	template <typename F>
	__device__ __forceinline__ void operator()(const WaveLds &lds, const RowSpace &sp, float qa2, int n, F before_loads
	                                           VSS_WC_ARG) const {
		if constexpr (LATE) {
			if (n <= 0)
				before_loads();
			wave_distances<MT, NCH, R>(sp, lds.q, qa2, lds.ids, n, lds.dist, before_loads);
		} else {
			before_loads();
			wave_distances<MT, NCH, R>(sp, lds.q, qa2, lds.ids, n, lds.dist);
		}
	}
};

// ---------------------------------------------------------------------------------------------------------
// Teams (the solo shape with T > 1 waves per query; k_search_solo<.., T>).  Measured on the one-query probe: even with every
// row already in L2 the scoring phase of the one-wave shape takes 2.4k cycles per expansion — the floor set by the ~400
// instructions ONE wave has to issue for an expansion's rows (pointers, 16 float4 loads per lane, FMAs, the transposed
// reduction); cold rows add ~1.2k on top.  A team puts that work on the compute unit's other SIMDs: wave 0 walks the graph exactly as before (it
// alone owns the candidate list, the visited set and every decision); waves 1 .. T-1 wait at a workgroup barrier, score a
// contiguous share of the rows wave 0 gathered, and meet it at a second barrier.  Two s_barriers per expansion instead of
// the engine's mailbox polling (that exchange has to serve several walkers; here there is one).  A row is still reduced
// by one lane group in wave order, so distances keep their bits whichever wave scores them.
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t CREW_ON = 1u, CREW_TOUCH = 2u, CREW_SPARE_SIMD = 4u, CREW_NO_REQUESTS = 8u; // SearchArgs::crew
constexpr uint32_t TOUCH_LISTS = 0x100u; // WaveLds::touch_lines / SearchArgs::touch_lines: bits 0-7 row lines, bit 8 ListTouch
struct TeamBox {
	int n;     // rows on offer (lds.ids[0..n)), < 0 = the walk is over
	float qa2; // the query's squared norm (cosine)
	uint32_t touch_n; // neighbour lists in `cells` whose rows the helpers pull into L2 after the second barrier (RowTouch)
	uint32_t pad;
	uint32_t cells[2][64];
};
constexpr uint32_t TEAM_BOX_BYTES = (uint32_t)sizeof(TeamBox); // the first bytes of a team's LDS (multiple of 16)
__device__ __forceinline__ void team_share(const RowSpace &sp, int n, int T, int wave, int &lo, int &hi) {
	const int RG = 64 >> sp.logG; // rows side by side in one register slot (a power of two): shares are multiples of it
	const int per = ((n + T - 1) / T + RG - 1) & ~(RG - 1); // (a mask, not `/ RG * RG`: RG is a run-time value — an integer division)
	lo = wave * per < n ? wave * per : n;
	hi = lo + per < n ? lo + per : n;
}
struct NoListCache {};
template <int MT, int NCH, int R, int T>
struct TeamScorer {
	// the walking wave touches the lists of the rows it accepts (ListTouch: one load); the rows of the cached lists are
	// touched by the helpers, which idle through the accept phase anyway (the walker only hands them the cells)
	static constexpr bool touches_rows = false, touches_lists = true, helpers_touch = true;
	__device__ __forceinline__ bool latency_mode() const {
		return false;
	}
	__device__ __forceinline__ void at_level(int) const {
	}
	TeamBox *box; // LDS
	// `cache` (level search): the list cache whose freshly arrived lists name the rows to touch; `touch` = touching is on
	template <typename F, class Cache>
	__device__ __forceinline__ void run(const WaveLds &lds, const RowSpace &sp, float qa2, int n, F before_loads, Cache &cache,
	                                    bool touch) const {
		if (n <= 0) {
			before_loads();
			return;
		}
		if (lane_id() == 0)
			box->n = n, box->qa2 = qa2;
		__syncthreads(); // the ids (and, per query, the staged query) are in LDS: the helpers start
		int lo, hi;
		team_share(sp, n, T, 0, lo, hi);
		wave_distances<MT, NCH, R>(sp, lds.q, qa2, lds.ids, hi, lds.dist, before_loads);
		uint32_t k = 0;
		if constexpr (!std::is_same<Cache, NoListCache>::value) {
			if (touch) { // (the lists requested in the shadow of the rows have arrived with them)
#pragma unroll
				for (int i = 0; i < Cache::slots; ++i)
					if (cache.fresh_bits() & (1u << i)) {
						box->cells[k & 1u][lane_id()] = cache.cells_of(i);
						++k;
					}
				cache.clear_fresh();
			}
		}
		if (lane_id() == 0)
			box->touch_n = k;
		__syncthreads(); // every share's distances are in LDS
	}
	template <typename F>
	__device__ __forceinline__ void operator()(const WaveLds &lds, const RowSpace &sp, float qa2, int n, F before_loads
	                                           VSS_WC_ARG) const {
		NoListCache none;
		run(lds, sp, qa2, n, before_loads, none, false);
	}
};
// rows of at most this many 128-byte lines are touched ahead by a team's helpers (a full lane group of NCH chunks per lane
// is NCH KiB; the looping kernels touch rows up to 1 KiB)
__host__ __device__ constexpr int team_touch_max_lines(int nch) {
	return nch > 0 ? 8 * nch : 8;
}
template <int MT, int NCH, int R, int T>
__device__ __forceinline__ void team_help(const WaveLds &lds, const RowSpace &sp, const TeamBox *box, int wave, uint32_t lines) {
	constexpr int LPH = (team_touch_max_lines(NCH) + T - 2) / (T - 1); // lines of a row one helper touches (T - 1 helpers)
	uint32_t sink[2][LPH];
#pragma unroll
	for (int k = 0; k < 2; ++k)
#pragma unroll
		for (int j = 0; j < LPH; ++j)
			sink[k][j] = 0;
	for (;;) {
		__syncthreads();
		const unsigned long long nq = *reinterpret_cast<const unsigned long long *>(box); // {n, qa2}: one LDS round trip
		const int n = uniform((int)(uint32_t)nq);
		if (n < 0)
			break;
		const float qa2 = __int_as_float(uniform((int)(uint32_t)(nq >> 32)));
		int lo, hi;
		team_share(sp, n, T, wave, lo, hi);
		if (hi > lo)
			wave_distances<MT, NCH, R>(sp, lds.q, qa2, lds.ids + lo, hi - lo, lds.dist + lo);
		__syncthreads();
		// RowTouch by the helpers: one dword of every 128-byte line of the rows the walker's freshly cached lists name
		const uint32_t tn = (uint32_t)uniform((int)box->touch_n);
		if (tn) {
#pragma unroll
			for (int k = 0; k < 2; ++k)
#pragma unroll
				for (int j = 0; j < LPH; ++j)
					asm volatile("" ::"v"(sink[k][j])); // the previous touches: landed an expansion ago
#pragma unroll
			for (int k = 0; k < 2; ++k) {
				if ((uint32_t)k >= tn)
					continue;
				const uint32_t id = box->cells[k][lane_id()];
				if (id == EMPTY_SLOT)
					continue;
				const char *row = reinterpret_cast<const char *>(sp.vectors + (size_t)id * sp.V);
#pragma unroll
				for (int j = 0; j < LPH; ++j) {
					const uint32_t l = (uint32_t)(wave - 1) + (uint32_t)j * (T - 1);
					if (l < lines)
						sink[k][j] = *reinterpret_cast<const uint32_t *>(row + l * 128u);
				}
			}
		}
	}
#pragma unroll
	for (int k = 0; k < 2; ++k)
#pragma unroll
		for (int j = 0; j < LPH; ++j)
			asm volatile("" ::"v"(sink[k][j]));
}

// The search engine's job exchange (LDS).  One mailbox per walking wave.  `ticket` packs {rows of the open job (high
// word), next unclaimed row (low word)}: a scoring wave claims a chunk with ONE returning 64-bit atomic add, so the
// snapshot it gets back — job size and chunk start — is consistent whatever the walker does meanwhile; a claim beyond
// the job size claims nothing.  The walker opens a job with one 64-bit RELEASE store after the ids (and, per query, the
// staged query and its norm) are in LDS; a claim ACQUIRES; `done` counts scored rows (release add by the scorer, acquire
// load by the waiting walker).
#ifdef VSS_PARANOID
#define VSS_TRACE(sp, idx, expr) ((sp).debug[idx] = (expr))
#define VSS_TRACE_INC(sp, idx) atomicAdd(lane_id() == 0 ? &(sp).debug[idx] : &(sp).debug[64 + lane_id()], 1u)
#else
#define VSS_TRACE(sp, idx, expr)
#define VSS_TRACE_INC(sp, idx)
#endif
// pauses between polls (units of 64 cycles; A/B builds): a walker waiting for its scores, a scoring wave that found no job
#ifndef VSS_WALKER_WAIT_SLEEP
#define VSS_WALKER_WAIT_SLEEP 1
#endif
#ifndef VSS_SCORER_IDLE_SLEEP
#define VSS_SCORER_IDLE_SLEEP 2
#endif
struct Mailbox {
	unsigned long long ticket;
	uint32_t done;
	uint32_t qa2_bits; // |query|^2 (cosine), as bits
	uint32_t slots; // row slots per claim of the open job: 1, 2 or R (a claim covers slots x (64 / G) rows)
	uint32_t pad[3];
};

// The mailboxes are addressed as LDS (address space 3) explicitly: ds_* instructions instead of flat ones.
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) float lds_f32;
typedef uint32_t lds_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) lds_u32x4 lds_u32x4_as3;
// (generic -> LDS through the integer value — the low 32 bits of a generic LDS address are the LDS offset — rather than an
// addrspacecast: hipcc of ROCm 7.2 lowers the cast's null check to an instruction its own verifier rejects)
#define VSS_LDS_PTR(type, ptr) ((type *)(uint32_t)(uintptr_t)(ptr))
#define VSS_LDS_LOAD(type, ptr) __hip_atomic_load(VSS_LDS_PTR(type, ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define VSS_LDS_STORE(type, ptr, v) __hip_atomic_store(VSS_LDS_PTR(type, ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define VSS_LDS_ADD(type, ptr, v) __hip_atomic_fetch_add(VSS_LDS_PTR(type, ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
// The hand-over points of the exchange carry release / acquire semantics (workgroup scope: an s_waitcnt around the LDS
// instruction, no cache maintenance), so that the payload — ids, distances, the staged query — is ordered against the
// ticket / done words by the memory model and not merely by the in-order LDS pipeline.  -DVSS_RELAXED_MAILBOX rebuilds the
// round-2 all-relaxed exchange (A/B).
#ifdef VSS_RELAXED_MAILBOX
#define VSS_MB_RELEASE __ATOMIC_RELAXED
#define VSS_MB_ACQUIRE __ATOMIC_RELAXED
#else
#define VSS_MB_RELEASE __ATOMIC_RELEASE
#define VSS_MB_ACQUIRE __ATOMIC_ACQUIRE
#endif
#define VSS_LDS_LOAD_ACQ(type, ptr) __hip_atomic_load(VSS_LDS_PTR(type, ptr), VSS_MB_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)
#define VSS_LDS_STORE_REL(type, ptr, v) __hip_atomic_store(VSS_LDS_PTR(type, ptr), (v), VSS_MB_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP)
#define VSS_LDS_ADD_ACQ(type, ptr, v) __hip_atomic_fetch_add(VSS_LDS_PTR(type, ptr), (v), VSS_MB_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)
#define VSS_LDS_ADD_REL(type, ptr, v) __hip_atomic_fetch_add(VSS_LDS_PTR(type, ptr), (v), VSS_MB_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP)

// No `if (lane == 0)` around the atomics of this exchange: with `x = 0; if (lane == 0) x = atomic(); x = readfirstlane(x);`
// inside a loop, hipcc (ROCm 7.2) threads the lane-0 branches on either side of the back edge together, folds
// readfirstlane of the constant on the other lanes' path, and those lanes leave the loop after the first round (found on the
// GPU with tools/microbench/mailbox_test).  Instead EVERY lane executes the atomic — lane 0 on the real word, lane i on
// cell i of a scrap area nobody reads (distinct addresses: one LDS instruction, nothing for the compiler's wave-level
// atomic combiner to rewrite) — and lane 0's return value is taken.
template <int MT, int NCH, int R>
__device__ __forceinline__ bool pool_score(Mailbox *mb, unsigned long long *scrap, const RowSpace &sp, const float4 *q,
                                           const uint32_t *ids, float *dist) {
	const int lane = lane_id();
	unsigned long long *ticket_or_scrap = lane == 0 ? &mb->ticket : scrap + lane;
	uint32_t *done_or_scrap = lane == 0 ? &mb->done : reinterpret_cast<uint32_t *>(scrap + lane);
	bool worked = false;
	for (;;) {
		// rows per claim: the walker's choice for the open job (fewer rows per scoring wave when many of them are idle:
		// a wave's latency grows by ~60 cycles per KiB it loads).  A stale value only changes how many rows this claim takes.
		uint32_t slots = (uint32_t)uniform((int)VSS_LDS_LOAD(lds_u32, &mb->slots));
		slots = slots == 1 || slots == 2 ? slots : (uint32_t)R;
		const uint32_t pass = slots * (64u >> sp.logG);
		const unsigned long long t = VSS_LDS_ADD_ACQ(lds_u64, ticket_or_scrap, (unsigned long long)pass); // claim: acquire
		const uint32_t c = (uint32_t)uniform((int)(uint32_t)t), n = (uint32_t)uniform((int)(uint32_t)(t >> 32));
		VSS_TRACE_INC(sp, 25);
		VSS_TRACE(sp, 27, c);
		VSS_TRACE(sp, 28, n);
		if (c >= n)
			break;
		VSS_TRACE_INC(sp, 26);
		const uint32_t cnt = n - c < pass ? n - c : pass;
#ifdef VSS_PARANOID // debug builds: never follow an id that cannot be a slot; leave a note instead
		{
			bool bad = n > 4096;
			for (uint32_t j = lane_id(); !bad && j < cnt; j += 64)
				bad = ids[c + j] >= sp.debug_rows;
			if (sp.debug && __ballot(bad)) {
				if (lane_id() == 0 && atomicAdd(&sp.debug[0], 1u) == 0) {
					sp.debug[1] = c, sp.debug[2] = n, sp.debug[3] = ids[c], sp.debug[4] = blockIdx.x;
					sp.debug[5] = (uint32_t)(threadIdx.x >> 6), sp.debug[6] = (uint32_t)(mb->ticket >> 32);
					sp.debug[7] = (uint32_t)mb->ticket, sp.debug[8] = cnt, sp.debug[9] = ids[c + cnt - 1];
				}
				VSS_LDS_ADD(lds_u32, done_or_scrap, cnt);
				worked = true;
				continue;
			}
		}
#endif
		const float qa2 = __uint_as_float(VSS_LDS_LOAD(lds_u32, &mb->qa2_bits));
		// (every variant reduces a row with the same lanes in the same order: same bits; all end with wave_sync)
		if (slots == 1)
			wave_distances<MT, NCH, 1>(sp, q, qa2, ids + c, (int)cnt, dist + c);
		else if (slots == 2)
			wave_distances<MT, NCH, 2>(sp, q, qa2, ids + c, (int)cnt, dist + c);
		else
			wave_distances<MT, NCH, R>(sp, q, qa2, ids + c, (int)cnt, dist + c);
		VSS_LDS_ADD_REL(lds_u32, done_or_scrap, cnt); // the distances above are published with this add: release
		VSS_TRACE_INC(sp, 29);
		worked = true;
	}
	return worked;
}

// LDS header of the search engine (k_search)
constexpr uint32_t ENGINE_MAX_WALKERS = 8; // (the host keeps to 4 unless told otherwise: vss_engine.hip search_walkers_cap)
// {exit flag, walkers left, pad} + crew box + mailboxes + 64 scrap cells (the dummy targets of pool_score's all-lane atomics)
constexpr uint32_t ENGINE_BOXES = 2 * ENGINE_MAX_WALKERS; // two job buffers (and mailboxes) per walker
constexpr uint32_t ENGINE_CREW_OFFSET = 16;                       // the crew box (16 bytes) follows {exit flag, walkers left, pad}
constexpr uint32_t ENGINE_SIMD_OFFSET = 32;                       // which SIMD each of the (at most 16) waves runs on, one byte each
constexpr uint32_t ENGINE_BOX_OFFSET = 48;                        // the mailboxes
constexpr uint32_t ENGINE_SCRAP_OFFSET = ENGINE_BOX_OFFSET + ENGINE_BOXES * 32;
constexpr uint32_t ENGINE_HEADER_BYTES = ENGINE_SCRAP_OFFSET + 64 * 8;

constexpr uint32_t POOL_SPIN_LIMIT = 1u << 26; // polls before a waiting wave gives up and traps (never hang the GPU)

// ---------------------------------------------------------------------------------------------------------
// Crew mode (round 4): the hand-over for a workgroup with ONE walker left.  The mailbox exchange above has to serve several
// walkers — scoring waves poll, claim with atomics, walkers poll back — and its polling waves share the walker's SIMD.  A
// lone walker (a launch of at most one query per compute unit: the single-query probe and the join chunks at wide rows, or
// the drain of a large launch once a workgroup's other walkers have found the queue dry) needs none of that: it writes the
// job size into the crew box and meets the scoring waves at a workgroup barrier; every scoring wave takes a fixed share of
// the rows, scores it, and all meet again at a second barrier.  Between expansions the scoring waves are parked at the
// barrier (no issue slots taken from the walker).  Two s_barriers per expansion, no atomics, no polling.  A row is reduced by
// one lane group in wave order whichever wave holds it: ids, distance bits and counters are what they were.
//   * The switch is one-way and safe: `walkers_left` only falls, a walker that has left never comes back (s_barrier counts
//     the waves of the workgroup that have not ended), and the walker raises `on` only between two of its own jobs, when no
//     row of it is in a scoring wave's hands.  Scoring waves notice `on` in their polling loop and move to crew_help().
//   * The walker dismisses the crew with n = -1 when the queue is dry.
//   (Round 6 measured the crew also pulling the ROWS of the lists its walker holds one expansion ahead into L2, for launches of
//   a handful of queries: 381 -> 424 us per one-query probe at 3M x 768 — the hand-over of the cells costs the walker 650 ticks
//   per expansion and the scoring waves' row time does not move: profiles/r06e_crew_probe_3m768_crew_row_touches_not_kept.txt.)
// ---------------------------------------------------------------------------------------------------------
struct CrewBox {
	int n;           // rows on offer in the walker's job buffer 0, < 0 = the walk is over
	float qa2;       // the query's squared norm (cosine)
	uint32_t walker; // bits 0-7: 2 x the walker slot (EngineSlot: staged query, ids, distances) + the job buffer (0 / 1) the rows
	                 // are in; bits 8+: the graph level the rows were gathered on (what the scoring waves touch ahead: CrewTouch)
	uint32_t on;     // non-zero: the workgroup is in crew mode
};
static_assert(sizeof(CrewBox) == 16 && sizeof(Mailbox) == 32, "engine LDS header layout");

template <int MT, int NCH, int R>
struct PoolScorer {
	static constexpr bool touches_rows = false, touches_lists = true, helpers_touch = false;
	Mailbox *mb;            // this walker's two mailboxes (job buffers 0 and 1)
	uint32_t *exit_flag;    // LDS: non-zero = the scoring waves are leaving
	uint32_t *engine_error; // pinned host word: set when a walker gave up waiting
	uint32_t *walkers_left; // LDS: walkers of this workgroup that still have queries
	uint32_t scorers;       // scoring waves of this workgroup
	CrewBox *crew;          // LDS
	uint32_t my_slot;       // this walker's slot
	uint32_t crew_ok;       // the launch allows crew mode (host: SearchArgs::crew)
	uint32_t no_requests;   // a walker running a crew asks for no lists ahead of time (CREW_NO_REQUESTS)
	uint32_t touch_ok;      // the host allows list touches for this launch (CREW_TOUCH: vss_set_search_touch / list_cap <= 64)
	mutable uint32_t crew_on = 0; // wave-uniform: this walker is the last one and runs the crew
	mutable uint32_t level = 0;   // wave-uniform: the graph level of the rows handed over next (descend; 0 = the base level)
	__device__ __forceinline__ void at_level(int l) const {
		level = (uint32_t)l;
	}

	// look-ahead list requests: worth their instructions unless the crew's touches keep every candidate's list in L2 anyway
	__device__ __forceinline__ bool wants_requests() const {
		return !(crew_on && no_requests);
	}

	// ListTouch (level_search_impl) while the walker is alone: its expansions are a latency chain, not a bandwidth stream —
	// where the host allows touches at all (VSS_SEARCH_TOUCH_LISTS=0 has to be a clean off for A/B)
	__device__ __forceinline__ bool latency_mode() const {
		return crew_on != 0 && touch_ok != 0;
	}
	__device__ __forceinline__ uint32_t active_walkers() const {
		const uint32_t active = (uint32_t)uniform((int)VSS_LDS_LOAD(lds_u32, walkers_left));
		return active ? active : 1u;
	}
	// offer the n ids of job buffer `buf` (already in LDS) to the scoring waves
	__device__ __forceinline__ void post(int buf, const RowSpace &sp, int n) const {
		Mailbox *box = mb + buf;
		// rows per claim: spread the job over the scoring waves this walker can count on (mine = scorers / active walkers: all
		// of them once its neighbours have finished), never more than R row slots per wave.  Without the two run-time integer
		// divisions this used to cost on the walker's serial path (rounds 2-6): with x1 = the row slots the job needs,
		// ceil(x1 / mine) <= 1  <=>  active * x1 <= scorers, and <= 2  <=>  active * ceil(x1 / 2) <= scorers.
		const uint32_t lg = 6u - sp.logG;
		const uint32_t x1 = ((uint32_t)n + (1u << lg) - 1u) >> lg, x2 = (x1 + 1u) >> 1;
		const uint32_t a = active_walkers();
		// (every lane stores the same values: no lane-0 branch, see pool_score)
		VSS_LDS_STORE(lds_u32, &box->slots, a * x1 <= scorers ? 1u : a * x2 <= scorers ? 2u : (uint32_t)R);
		VSS_LDS_STORE(lds_u32, &box->done, 0u);
		// one 64-bit atomic store opens the job: {n rows, next row 0} — release: ids, query and the words above come first
		VSS_LDS_STORE_REL(lds_u64, &box->ticket, (unsigned long long)(uint32_t)n << 32);
		VSS_TRACE_INC(sp, 16);
		VSS_TRACE(sp, 17, (uint32_t)n);
	}
	// block until the n rows of job buffer `buf` have been scored.  The walker does not score: it keeps the candidate
	// list in registers, and a scoring pass on top of that would not fit the 128 registers a 1024-thread workgroup
	// allows (the engine always runs at least one scoring wave).
	__device__ __forceinline__ void wait(int buf, const RowSpace &sp, int n) const {
		Mailbox *box = mb + buf;
		uint32_t spins = 0;
		while (uniform((int)VSS_LDS_LOAD_ACQ(lds_u32, &box->done)) < n) { // acquire: pairs with the scorers' release add
			__builtin_amdgcn_s_sleep(VSS_WALKER_WAIT_SLEEP);
			if ((spins & 1023u) == 0) {
				VSS_TRACE(sp, 20, spins);
				VSS_TRACE(sp, 21, VSS_LDS_LOAD(lds_u32, &box->done));
			}
			// Never hang the GPU: a wait that cannot end (it never should) raises the workgroup's exit flag — the scoring
			// waves leave, the other walkers leave from their own waits — and reports through *engine_error.
			if (++spins > POOL_SPIN_LIMIT || ((spins & 1023u) == 0 && uniform((int)VSS_LDS_LOAD(lds_u32, exit_flag)))) {
				if (lane_id() == 0) {
					VSS_LDS_STORE(lds_u32, exit_flag, 2u);
					__hip_atomic_store(engine_error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				}
				__builtin_amdgcn_endpgm();
			}
		}
		wave_sync();
		VSS_TRACE_INC(sp, 18);
	}
	// Hand the n ids of job buffer `buf` (already in LDS) to the scoring waves — through the mailbox, or, for the last walker
	// of the workgroup, behind the crew's first barrier.  The switch to crew mode happens here, between two jobs.
	__device__ __forceinline__ void begin(int buf, const RowSpace &sp, float qa2, int n) const {
		if (crew_ok && !crew_on && uniform((int)VSS_LDS_LOAD(lds_u32, walkers_left)) == 1) {
			crew_on = 1u; // (every lane stores the same value)
#ifdef VSS_PHASE_TIMERS
			if (lane_id() < 3) // the first scoring wave's tick accumulators (the scrap cells of the mailbox atomics, idle from now on)
				reinterpret_cast<unsigned long long *>(reinterpret_cast<unsigned char *>(crew) - ENGINE_CREW_OFFSET + ENGINE_SCRAP_OFFSET)[lane_id()] = 0;
#endif
			VSS_LDS_STORE_REL(lds_u32, &crew->on, 1u);
		}
		if (crew_on) {
			if (lane_id() == 0) {
				crew->n = n;
				crew->qa2 = qa2;
				crew->walker = 2u * my_slot + (uint32_t)buf + (level << 8);
			}
			lds_barrier(); // the ids (and, per query, the staged query) are in LDS: the crew starts
			return;
		}
		post(buf, sp, n);
	}
	// block until the rows handed over by begin(buf, .., n) have their distances in LDS
	__device__ __forceinline__ void end(int buf, const RowSpace &sp, int n) const {
		if (crew_on)
			lds_barrier(); // every share's distances are in LDS (this wave's own loads — the look-ahead — stay in flight)
		else
			wait(buf, sp, n);
	}
	// Scorer interface (descend, level_search_impl): job buffer 0
	template <typename F>
	__device__ __forceinline__ void operator()(const WaveLds &lds, const RowSpace &sp, float qa2, int n, F before_loads
	                                           VSS_WC_ARG) const {
		if (n <= 0) {
			before_loads();
			return;
		}
		VSS_TICK(tp0);
		begin(0, sp, qa2, n);
		VSS_TICK(tpb);
		VSS_ACC(t_solo_passes, tp0, tpb);
		before_loads();
		VSS_TICK(tp1);
		end(0, sp, n);
		VSS_TICK(tp3);
		VSS_ACC(t_look, tp0, tp1);
		VSS_ACC(t_sync2, tp1, tp3);
	}
	// the walker has found the queue dry
	__device__ __forceinline__ void dismiss_crew() const {
		if (lane_id() == 0)
			crew->n = -1;
		lds_barrier();
	}
};

// ---------------------------------------------------------------------------------------------------------
// search_for_one_: greedy descent from (closest) through levels begin_level .. end_level+1.
// AHEAD (the search kernels; round 6): a step of the descent is three dependent memory round trips — upper_off[closest], the
// list, the rows it names — and the walker only waits while the rows are scored.  In that shadow it asks for the list the
// descent reads next IF no row turns out closer (once per level, always): the same node's list one level down.  The other
// case — a row IS closer — is served by the scoring waves of a crew, which pull upper_off[] and the list line of the rows
// they score into L2 (CrewTouch, `level`).  Pure latency hiding: which lists are read, in which order, does not change.
template <int MT, bool AHEAD = false, class Scorer>
__device__ __forceinline__ uint32_t descend(const GraphView &gv, WaveLds &lds, float qa2, uint32_t closest,
                                            int begin_level, int end_level, const Scorer &score, WorkCounters &wc) {
	const int lane = lane_id();
	float closest_dist = wave_distance_one<MT>(gv.sp, lds.q, qa2, closest);
	wc.distances += 1;
	ListPrefetch below; // AHEAD: the list of below.slot at below_level, one cell per lane (lists of at most 64 cells)
	int below_level = -1;
	const bool can_ahead = AHEAD && gv.M0 <= 64;
	for (int level = begin_level; level > end_level; --level) {
		bool changed;
		do {
			changed = false;
			const bool have_first = can_ahead && below.slot == closest && below_level == level;
			const int n = gather_neighbors<false>(gv, lds, closest, level, have_first, have_first ? below.masked() : EMPTY_SLOT);
			if constexpr (AHEAD)
				score.at_level(level);
			score(lds, gv.sp, qa2, n, [&] {
				if constexpr (AHEAD) {
					if (can_ahead && !(below.slot == closest && below_level == level - 1)) {
						below.slot = EMPTY_SLOT;
						below.request(gv, closest, level - 1); // (level 0: the list the base level's first expansion reads)
						below_level = level - 1;
					}
				}
			} VSS_WC_PASS);
			wc.distances += n;
			wc.cycles += 1;
			// first occurrence of the minimum, taken only if strictly smaller (index.hpp:3835-3842)
			for (int off = 0; off < n; off += 64) {
				const float d = (off + lane < n) ? lds.dist[off + lane] : __builtin_inff();
				float m = d;
				m = fminf(m, lane_xor<32>(m));
				m = fminf(m, lane_xor<16>(m));
				m = fminf(m, lane_xor<8>(m));
				m = fminf(m, lane_xor<4>(m));
				m = fminf(m, lane_xor<2>(m));
				m = fminf(m, lane_xor<1>(m));
				if (m < closest_dist) {
					const unsigned long long who = __ballot(d == m);
					closest_dist = m;
					closest = lds.ids[off + __builtin_ctzll(who)];
					changed = true;
				}
			}
			wave_sync();
		} while (changed);
	}
	if constexpr (AHEAD)
		score.at_level(end_level);
	return closest;
}

// ---------------------------------------------------------------------------------------------------------
// search_to_insert_ (INSERT) / search_to_find_in_base_ (!INSERT) on one level.
//   L: the candidate list (one sorted list with "expanded" marks; see oracle header for the equivalence with
//      the reference's heap + sorted buffer): WaveList<E> in registers, or MemList for limits beyond 64 * MAX_LIST_REGS.
//   TOMB (search only: the index holds tombstones, or a predicate filters the result): L holds the admitted rows only —
//      it is the reference's `top` and defines the radius — and every accepted candidate waits in the queue `cq` (the
//      reference's unbounded `next` heap).
// Returns LEVEL_OK, or why the query has to be re-run with more scratch.
enum { LEVEL_OK = 0, LEVEL_VISITED_OVERFLOW = 1, LEVEL_QUEUE_OVERFLOW = 2, LEVEL_INTERNAL = 3 /* never: the host fails loudly */,
       LEVEL_OK_RETRIED = 16 /* status word only: answered, after the walker repeated the search over its visited set in HBM */ };

// one list in flight: exactly the round-2 ListPrefetch (no replacement state)
template <>
struct ListCache<1> {
	static constexpr int slots = 1;
	ListPrefetch one;
	__device__ __forceinline__ bool find(uint32_t want, uint32_t &out) const {
		out = one.masked();
		return one.slot == want;
	}
	__device__ __forceinline__ void request(const GraphView &gv, uint32_t want, int level) {
		one.request(gv, want, level);
	}
	__device__ __forceinline__ uint32_t cells_of(int) const {
		return one.masked();
	}
	__device__ __forceinline__ uint32_t fresh_bits() const {
		return one.fresh;
	}
	__device__ __forceinline__ void clear_fresh() {
		one.fresh = 0;
	}
};

// ---------------------------------------------------------------------------------------------------------
// RowTouch (solo search kernel, launches of at most one query per compute unit = latency, not bandwidth): the rows an expansion
// scores are named by a list that, six times in ten, sits in the list cache one expansion EARLIER (ListCache: the best two unexpanded entries).  As soon as
// such a list has arrived — at the start of the accept phase, whose ~1.5k cycles are pure wave-local bookkeeping — the wave
// touches one dword of every 128-byte line of the rows it names.  The values are never used; the lines are in this XCD's
// L2 when the expansion that scores them asks (≈200 instead of ≈900 cycles, MI355X_MICROARCH.md).  Which rows are scored,
// their distances and every counter stay what they were: the loads only move cache lines.  The price is bandwidth (all the
// rows of both lists, visited or not), so the host turns it on only for launches that cannot be bound by it.
// The touched dwords stay in `v` until retire() — one expansion later, long after they have landed.
// In a team the helpers do this (team_help); this struct is the lone wave's version (rows of at most four lines).
// ---------------------------------------------------------------------------------------------------------
template <int K, int LINES>
struct RowTouch {
	uint32_t v[K][LINES];
	__device__ __forceinline__ RowTouch() {
#pragma unroll
		for (int i = 0; i < K; ++i)
#pragma unroll
			for (int l = 0; l < LINES; ++l)
				v[i][l] = 0;
	}
	template <class Cache>
	__device__ __forceinline__ void issue(Cache &cache, const RowSpace &sp, uint32_t lines) {
#pragma unroll
		for (int i = 0; i < K; ++i) {
			if (!(cache.fresh_bits() & (1u << i))) // wave-uniform
				continue;
			const uint32_t id = cache.cells_of(i);
			if (id != EMPTY_SLOT) { // (rows of fewer lines touch their last line again: one predicate, no branch per line)
				const char *row = reinterpret_cast<const char *>(sp.vectors + (size_t)id * sp.V);
#pragma unroll
				for (int l = 0; l < LINES; ++l)
					v[i][l] = *reinterpret_cast<const uint32_t *>(row + ((uint32_t)l < lines ? (uint32_t)l : lines - 1) * 128u);
			}
		}
		cache.clear_fresh();
	}
	__device__ __forceinline__ void retire() {
#pragma unroll
		for (int i = 0; i < K; ++i)
#pragma unroll
			for (int l = 0; l < LINES; ++l)
				asm volatile("" ::"v"(v[i][l]));
	}
};

// PK: neighbour lists kept in flight (0 = the list type's default); MERGE_REGS: widest register list whose accepted
// candidates are merged in batches (the 1024-thread search engine cannot afford the temporaries of an 8-register merge)
template <int MT, bool INSERT, bool TOMB, int PK = 0, int MERGE_REGS = MAX_LIST_REGS, class List, class Queue, class Scorer>
__device__ __forceinline__ int level_search_impl(const GraphView &gv, WaveLds &lds, float qa2, uint32_t start,
                                                 uint32_t new_slot, int level, int limit, List &L, Queue &cq,
                                                 const Scorer &score, WorkCounters &wc) {
	const int lane = lane_id();
	lds.visited.clear();
	L.reset(limit);
	if (lane == 0)
		lds.visited.test_and_set(start);
	lds.visited.count = 1;
	const float d0 = wave_distance_one<MT>(gv.sp, lds.q, qa2, start);
	wc.distances += 1;
	wave_sync();
	float radius = d0;
	if (TOMB) {
		cq.restart();
		cq.push(d0, start, 0.f, false);
		if (gv.admitted(start))
			L.template insert<INSERT>(d0, start);
	} else {
		L.template insert<INSERT>(d0, start);
	}

	constexpr int SLOTS = PK > 0 ? PK : List::prefetch_slots; // neighbour lists kept in flight
	ListCache<SLOTS> ahead;
	const bool can_prefetch = gv.list_cap(level) <= 64;
	constexpr bool TOUCH = Scorer::touches_rows && !INSERT;        // RowTouch by this wave
	constexpr bool TOUCH_L = Scorer::touches_lists && !INSERT;     // ListTouch by this wave
	constexpr bool TOUCH_H = Scorer::helpers_touch && !INSERT;     // RowTouch by the team's helpers
	RowTouch<(TOUCH ? SLOTS : 1), (TOUCH ? 4 : 1)> touch;
	uint32_t list_sink = 0;
	// the lists of the best two entries still unexpanded (the candidate queue's front when rejected rows are tracked)
	auto request_ahead = [&] {
		if (!can_prefetch)
			return;
		float nd;
		uint32_t ns;
		if constexpr (TOMB) {
			if (cq.empty())
				return;
			cq.front(nd, ns);
			ahead.request(gv, ns, level);
		} else {
			if constexpr (SLOTS > 1) { // (register lists only: both slot words in one pass over the list)
				uint32_t s2 = 0;
				const int have = L.first_two_unexpanded(ns, s2);
				if (have > 0)
					ahead.request(gv, ns, level);
				if (have > 1)
					ahead.request(gv, s2, level);
			} else {
				if (L.first_unexpanded_entry(nd, ns) >= 0)
					ahead.request(gv, ns & ~EXPANDED_BIT, level);
			}
		}
	};
	for (;;) {
		VSS_TICK(tk0);
		float cd;
		uint32_t cs;
		if (TOMB) {
			if (cq.empty())
				break;
			cq.front(cd, cs);
			if (L.size > 0 && cd > radius) // radius is unbounded until the first admitted entry (usearch: UB, Q6)
				break;
			cq.pop();
		} else {
			const int pos = L.first_unexpanded_entry(cd, cs);
			if (pos < 0)
				break;
			L.mark_expanded(pos);
		}
		wc.cycles += 1;
		if (INSERT && cs == new_slot)
			continue;
		VSS_TICK(tk1);
		VSS_ACC(t_pick, tk0, tk1);
		uint32_t first_cells;
		const bool have_first = can_prefetch && ahead.find(cs, first_cells);
		VSS_COUNT(t_look, have_first ? 1u : 0u); // (profiling builds: expansions whose list was in the cache)
		const int n = gather_neighbors<true>(gv, lds, cs, level, have_first, first_cells);
		VSS_TICK(tk2);
		VSS_ACC(t_gather, tk1, tk2);
		if (n < 0)
			return LEVEL_VISITED_OVERFLOW;
		auto look_ahead = request_ahead;
		if (n == 0) {
			look_ahead();
			continue;
		}
		if constexpr (TOUCH_H)
			score.run(lds, gv.sp, qa2, n, look_ahead, ahead, (lds.touch_lines & 0xFFu) && can_prefetch);
		else
			score(lds, gv.sp, qa2, n, look_ahead VSS_WC_PASS);
		wc.distances += n;
		VSS_TICK(tk3);
		VSS_ACC(t_dist, tk2, tk3);
		if constexpr (TOUCH) {
			if ((lds.touch_lines & 0xFFu) && can_prefetch) { // the lists requested in the shadow of these rows have arrived with them
				touch.retire();
				VSS_COUNT(t_slice, (unsigned)__builtin_popcount(ahead.fresh_bits())); // (profiling builds: lists touched)
				touch.issue(ahead, gv.sp, lds.touch_lines & 0xFFu);
			}
		}
		bool touch_lists = false;
		if constexpr (TOUCH_L) {
			asm volatile("" ::"v"(list_sink)); // last expansion's list touches: long landed
			touch_lists = (lds.touch_lines & TOUCH_LISTS) || score.latency_mode();
		}
		for (int off = 0; off < n; off += 64) {
			const bool have = off + lane < n;
			const float d = have ? lds.dist[off + lane] : 0.f;
			const uint32_t id = have ? lds.ids[off + lane] : 0;
			const uint32_t live = (TOMB && have) ? (gv.admitted(id) ? 1u : 0u) : 0u;
			unsigned long long pass = __ballot(have && (L.size < limit || d < radius));
			if constexpr (TOUCH_L) {
				// ListTouch: a row about to enter the candidate list will be asked for its neighbour list by the look-ahead of one
				// of the next expansions — a load the scoring phase ends up waiting for (the wave's loads retire in order).  Pull
				// the list's line into L2 now; the value is never used.
				if (touch_lists && off == 0 && have && (L.size < limit || d < radius))
					list_sink = *gv.list_ptr(id, level);
			}
			if constexpr (!TOMB) { // (round 6: the list's own accept loop — one basic block per candidate, see WaveList::accept)
				L.accept(d, id, pass, radius);
				continue;
			}
			while (pass) {
				const int j = __builtin_ctzll(pass);
				pass &= pass - 1;
				const float dj = read_lane(d, j);
				if (L.size < limit || dj < radius) {
					const uint32_t idj = read_lane(id, j);
					if (TOMB) {
						if (!cq.push(dj, idj, radius, L.size >= limit))
							return LEVEL_QUEUE_OVERFLOW;
						if (read_lane(live, j))
							L.template insert<INSERT>(dj, idj);
						if (L.size > 0)
							radius = L.last_distance();
					} else {
						L.template insert<INSERT>(dj, idj);
						radius = L.last_distance();
					}
				}
			}
		}
		wave_sync();
		VSS_TICK(tk4);
		VSS_ACC(t_accept, tk3, tk4);
	}
	if constexpr (TOUCH)
		touch.retire();
	if constexpr (TOUCH_L)
		asm volatile("" ::"v"(list_sink));
	return LEVEL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// search_to_find_in_base_ with ONE expansion of look-ahead (search engine only; no tombstones / predicate).
// An expansion is a chain: candidate -> its list -> the rows it names -> accept -> next candidate.  While the rows of the
// current candidate are being scored, the walker takes the best remaining unexpanded entry — the candidate that comes next
// unless one of the rows in flight beats it — filters ITS list through the visited set WITHOUT marking anything, and offers
// those rows to the scoring waves as well (second job buffer).  If that candidate is indeed expanded next, its rows are
// re-filtered for real (in list order, exactly what gather_neighbors would keep at that moment — anything visited since
// the probe drops out) and their distances are already there; if it is not, the speculative distances wait until it is
// (or are overwritten).  Distances are pure functions of (query, row): which lists are expanded, in which order, with
// which rows, and every counter, are those of level_search_impl.  What changes is that scoring overlaps the walker's
// bookkeeping; the price is rows scored for candidates that were evicted before their turn, so the walker only
// speculates while scoring waves are idle (`max_active`: walkers of the workgroup still running).
struct SpecBuffers { // (selected with ?: — an array indexed at run time would live in scratch memory)
	uint32_t *ids0, *ids1;
	float *dist0, *dist1;
	__device__ __forceinline__ uint32_t *ids(int b) const {
		return b ? ids1 : ids0;
	}
	__device__ __forceinline__ float *dist(int b) const {
		return b ? dist1 : dist0;
	}
};

template <int MT, class List, class Pool>
__device__ __forceinline__ int level_search_spec(const GraphView &gv, WaveLds &lds, const SpecBuffers &sb, float qa2,
                                                 uint32_t start, int limit, List &L, const Pool &pool, uint32_t max_active,
                                                 WorkCounters &wc) {
	const int lane = lane_id();
	lds.visited.clear();
	L.reset(limit);
	if (lane == 0)
		lds.visited.test_and_set(start);
	lds.visited.count = 1;
	const float d0 = wave_distance_one<MT>(gv.sp, lds.q, qa2, start);
	wc.distances += 1;
	wave_sync();
	float radius = d0;
	L.insert(d0, start);

	ListPrefetch ahead;
	bool have_spec = false;
	uint32_t spec_slot = EMPTY_SLOT;
	int spec_buf = 1, spec_n = 0;
	for (;;) {
		VSS_TICK(tk0);
		float cd;
		uint32_t cs;
		const int pos = L.first_unexpanded_entry(cd, cs);
		if (pos < 0)
			break;
		L.mark_expanded(pos);
		wc.cycles += 1;
		// the predicted successor: the best entry still unexpanded
		uint32_t ns = EMPTY_SLOT;
		{
			float nd;
			uint32_t nw;
			if (L.first_unexpanded_entry(nd, nw) >= 0)
				ns = nw;
		}
		VSS_TICK(tk1);
		VSS_ACC(t_pick, tk0, tk1);
		int b, n;
		if (have_spec && spec_slot == cs) {
			// its rows were offered ahead of time: keep those not visited since (list order is preserved)
			b = spec_buf;
			have_spec = false;
			if (ns != EMPTY_SLOT)
				ahead.request(gv, ns, 0);
			pool.wait(b, gv.sp, spec_n);
			n = 0;
			uint32_t bad = 0;
			for (int off = 0; off < spec_n; off += 64) {
				const bool have = off + lane < spec_n;
				const uint32_t id = have ? sb.ids(b)[off + lane] : EMPTY_SLOT;
				const float d = have ? sb.dist(b)[off + lane] : 0.f;
				const bool take = mark_first_visit(lds.visited, id, have, gv.twins != 0, bad);
				const unsigned long long m = __ballot(take);
				wave_sync(); // everything of this chunk is in registers before cells at or below it are rewritten
				if (take) {
					const int at = n + __popcll(m & lanes_below(lane));
					sb.ids(b)[at] = id;
					sb.dist(b)[at] = d;
				}
				n += __popcll(m);
			}
			lds.visited.count += n;
			wave_sync();
			if (lds.visited.count > lds.visited.limit || (lds.visited.compact && __ballot(bad != 0)))
				return LEVEL_VISITED_OVERFLOW;
		} else {
			b = have_spec ? 1 - spec_buf : 0;
			lds.ids = sb.ids(b);
			n = gather_neighbors<true>(gv, lds, cs, 0, ahead.slot == cs, ahead.masked());
			if (n < 0)
				return LEVEL_VISITED_OVERFLOW;
			if (n > 0)
				pool.post(b, gv.sp, n);
			if (ns != EMPTY_SLOT)
				ahead.request(gv, ns, 0);
		}
		VSS_TICK(tk2);
		VSS_ACC(t_gather, tk1, tk2);
		// look ahead: offer the successor's unvisited rows while this candidate's are being scored / accepted
		if (ns != EMPTY_SLOT && !(have_spec && spec_slot == ns) && pool.active_walkers() <= max_active) {
			const int ob = 1 - b;
			if (have_spec) // a guess that was overtaken: its rows must be out of the scoring waves' hands before reuse
				pool.wait(spec_buf, gv.sp, spec_n);
			have_spec = false;
			const uint32_t id = ahead.slot == ns ? ahead.masked() : EMPTY_SLOT; // one cell per lane (lists of <= 64 cells)
			const bool take = id != EMPTY_SLOT && !lds.visited.contains(id);
			const unsigned long long m = __ballot(take);
			if (take)
				sb.ids(ob)[__popcll(m & lanes_below(lane))] = id;
			const int n2 = __popcll(m);
			wave_sync();
			if (n2 > 0) {
				pool.post(ob, gv.sp, n2);
				have_spec = true;
				spec_slot = ns;
				spec_buf = ob;
				spec_n = n2;
			}
		}
		VSS_TICK(tk2b);
		VSS_ACC(t_look, tk2, tk2b);
		if (n == 0)
			continue;
		pool.wait(b, gv.sp, n); // (returns at once for rows that were offered ahead of time)
		wc.distances += n;
		VSS_TICK(tk3);
		VSS_ACC(t_dist, tk2b, tk3);
		for (int off = 0; off < n; off += 64) {
			const bool have = off + lane < n;
			const float d = have ? sb.dist(b)[off + lane] : 0.f;
			const uint32_t id = have ? sb.ids(b)[off + lane] : 0;
			const unsigned long long pass = __ballot(have && (L.size < limit || d < radius));
			L.accept(d, id, pass, radius);
		}
		wave_sync();
		VSS_TICK(tk4);
		VSS_ACC(t_accept, tk3, tk4);
	}
	if (have_spec) // nothing may be in flight on this walker's buffers when the next query starts
		pool.wait(spec_buf, gv.sp, spec_n);
	return LEVEL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// search_to_find_in_base_, software-pipelined (round 4; the workgroup engine: no tombstones / predicate, register lists,
// neighbour lists of at most 64 cells — everything else takes level_search_impl).
//
// An expansion is a chain: scores -> accept them into the candidate list -> the best unexpanded entry -> its list -> the
// unvisited rows it names -> their scores.  The walker does not score, so while the scoring waves fetch an expansion's rows
// it only waits (2.5-4k cycles at 768 dimensions), and while it runs the accept phase (~2k cycles of sorted inserts) the
// scoring waves idle.  The two are overlapped here, EXACTLY: which entry is expanded next can be told from the fresh
// scores without inserting anything —
//     m      = the smallest fresh distance that the radius test admits (list not full, or d < radius),
//     e      = the best entry of the list still unexpanded,
//     next   = the row of m if m exists and (no e, or m < e.distance);  else e;  else nobody (the search is over)
// because, absent exact ties with m: (1) m is admitted whatever the lane order of the inserts — every other fresh row is
// larger, so fewer than `limit` rows of list U fresh lie below m and the radius stays above it; (2) nothing inserted later
// evicts it (only the largest entry ever falls off); (3) if m > e.distance every admitted fresh row lands behind e and e is
// never the largest, so e survives.  So the walker picks `next` first, filters its list through the visited set and hands
// its rows over (marks set by a gather do not depend on the accept phase; their order is the sequential one), THEN accepts
// the previous expansion's rows in the shadow of those loads, and finally marks `next` — by then the first unexpanded entry
// of the list, which is checked (LEVEL_INTERNAL otherwise: the host refuses the answer).  Any exact tie of m (with another
// fresh row, with a list entry) or a NaN takes the plain order for that expansion: accept, then pick.  Ids, distance bits,
// the order of expansions and both work counters are those of level_search_impl.
//
// ListTouch moves to the scoring waves of a crew (CrewTouch): every row being scored has the lines of its own neighbour list
// pulled into L2 — the successor is usually one of them, and its list is asked for the moment the scores arrive.
template <int MT, int PK, class List, class Pool>
__device__ __forceinline__ int level_search_pipelined(const GraphView &gv, WaveLds &lds, const SpecBuffers &sb, float qa2,
                                                      uint32_t start, int limit, List &L, const Pool &pool, WorkCounters &wc) {
	const int lane = lane_id();
	lds.visited.clear();
	L.reset(limit);
	if (lane == 0)
		lds.visited.test_and_set(start);
	lds.visited.count = 1;
	const float d0 = wave_distance_one<MT>(gv.sp, lds.q, qa2, start);
	wc.distances += 1;
	wave_sync();
	float radius = d0;
	L.insert(d0, start);

	ListCache<PK> ahead;
	// The best entry still unexpanded, (e_d, e_s) — (+inf, EMPTY_SLOT) if there is none — found HERE, right after an expansion's
	// candidate has been marked, when the walker is about to wait for scores anyway, and used twice: now, to ask for the lists of
	// the best two such entries ahead of time, and by the next pick, which compares it with the smallest fresh distance (the
	// list does not change in between).  Round 6: the pick used to look for it again, on the critical path between the scores'
	// arrival and the successor's rows going out.
	float e_d = __builtin_inff();
	uint32_t e_s = EMPTY_SLOT;
	auto request_ahead = [&] {
		float d1 = 0.f;
		uint32_t s1 = 0, s2 = 0;
		int have;
		if constexpr (PK > 1) { // both slot words in one pass over the list
			have = L.first_two_unexpanded_entry(d1, s1, s2);
		} else {
			have = L.first_unexpanded_entry(d1, s1) >= 0 ? 1 : 0;
		}
		e_d = have > 0 ? d1 : __builtin_inff();
		e_s = have > 0 ? s1 : EMPTY_SLOT;
		if (!pool.wants_requests())
			return;
		if (have > 0)
			ahead.request(gv, s1, 0);
		if constexpr (PK > 1) {
			if (have > 1)
				ahead.request(gv, s2, 0);
		}
	};
	// filter the list of `cs` through the visited set into job buffer `buf` and hand the rows over; returns their number
	// (0: nothing to score, no job; < 0: visited-set overflow)
	auto open_expansion = [&](uint32_t cs, int buf) -> int {
		VSS_TICK(tg0);
		uint32_t first_cells;
		const bool have_first = ahead.find(cs, first_cells);
		VSS_COUNT(t_slice, have_first ? 1u : 0u); // (profiling builds: expansions whose list was in the cache)
		lds.ids = sb.ids(buf);
		const int n = gather_neighbors<true>(gv, lds, cs, 0, have_first, first_cells);
		VSS_TICK(tg1);
		VSS_ACC(t_gather, tg0, tg1);
		if (n > 0)
			pool.begin(buf, gv.sp, qa2, n);
		VSS_TICK(tg2);
		VSS_ACC(t_solo_passes, tg1, tg2);
		return n;
	};
	// sorted_buffer inserts of one expansion's rows (at most 64: one per lane), as level_search_impl does them
	auto accept = [&](int buf, int n) {
		const bool have = lane < n;
		const float d = have ? sb.dist(buf)[lane] : 0.f;
		const uint32_t id = have ? sb.ids(buf)[lane] : 0;
		const unsigned long long pass = __ballot(have && (L.size < limit || d < radius));
		L.accept(d, id, pass, radius);
	};

	int b = 0, n_cur = 0; // n_cur > 0: the rows of the candidate picked last are with the scoring waves, in job buffer b
	for (;;) {
		if (n_cur == 0) { // no scores pending: the plain order — pick the best unexpanded entry, open its expansion
			VSS_TICK(tk0);
			float cd;
			uint32_t cs;
			const int pos = L.first_unexpanded_entry(cd, cs);
			if (pos < 0)
				break;
			L.mark_expanded(pos);
			wc.cycles += 1;
			VSS_TICK(tk1);
			VSS_ACC(t_pick, tk0, tk1);
			n_cur = open_expansion(cs, b);
			if (n_cur < 0)
				return LEVEL_VISITED_OVERFLOW;
			request_ahead();
			continue;
		}
		VSS_TICK(tw0);
		pool.end(b, gv.sp, n_cur);
		VSS_TICK(tw1);
		VSS_ACC(t_sync2, tw0, tw1);
		wc.distances += n_cur;
		// ---- who is expanded next?  (see the header)
		uint32_t next = EMPTY_SLOT;
		bool tie;
		{
			const bool have = lane < n_cur;
			const float d = have ? sb.dist(b)[lane] : 0.f;
			const uint32_t id = have ? sb.ids(b)[lane] : 0;
			const bool admitted = have && (L.size < limit || d < radius);
			float m = admitted ? d : __builtin_inff();
			m = fminf(m, lane_xor<32>(m));
			m = fminf(m, lane_xor<16>(m));
			m = fminf(m, lane_xor<8>(m));
			m = fminf(m, lane_xor<4>(m));
			m = fminf(m, lane_xor<2>(m));
			m = fminf(m, lane_xor<1>(m));
			// `!(d > m)`: the rows at the minimum AND any NaN.  Ties that decide an order by position: two fresh rows at the
			// minimum (or a NaN next to it); a minimum that is not finite (+-inf; +inf also stands for a lone NaN — the padding's
			// values); the minimum EQUAL to the best unexpanded entry's distance (new goes before equal).  A tie with an entry
			// that is already expanded decides nothing: every entry in front of e is expanded, m lands in front of its equal,
			// and "the first unexpanded entry" is the same either way.  (e_d, e_s): found by request_ahead() before the wait.
			const unsigned long long who = __ballot(admitted && !(d > m));
			const float ms = __int_as_float(uniform(__float_as_int(m))); // (every lane holds it: a scalar for the branches below)
			tie = who && (__popcll(who) > 1 || !(__builtin_fabsf(ms) < __builtin_inff()) || ms == e_d);
			if (!tie)
				next = (who && ms < e_d) ? read_lane(id, __builtin_ctzll(who)) : e_s;
		}
		VSS_TICK(tw2);
		VSS_ACC(t_pick, tw1, tw2);
		if (tie) { // an exact tie decides by position: accept first, then pick the plain way
			accept(b, n_cur);
			n_cur = 0;
			VSS_TICK(tw3);
			VSS_ACC(t_accept, tw2, tw3);
			continue;
		}
		// ---- the successor's rows go out first ...
		const int nb = 1 - b;
		int n_next = 0;
		if (next != EMPTY_SLOT) {
			wc.cycles += 1;
			n_next = open_expansion(next, nb);
			if (n_next < 0)
				return LEVEL_VISITED_OVERFLOW; // (no job in flight: the failed gather hands nothing over)
		}
		// ---- ... and the scores that just arrived are accepted in the shadow of those loads
		VSS_TICK(ta0);
		accept(b, n_cur);
		uint32_t first = EMPTY_SLOT;
		int pos;
		{
			float cd;
			uint32_t w = EMPTY_SLOT;
			pos = L.first_unexpanded_entry(cd, w);
			if (pos >= 0)
				first = w;
		}
		if (first != next) { // cannot happen (header); never leave a job in flight behind, then let the host refuse the answer
			if (n_next > 0)
				pool.end(nb, gv.sp, n_next);
			return LEVEL_INTERNAL;
		}
		if (next == EMPTY_SLOT)
			break;
		L.mark_expanded(pos);
		request_ahead();
		VSS_TICK(ta1);
		VSS_ACC(t_accept, ta0, ta1);
		b = nb;
		n_cur = n_next;
	}
	return LEVEL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// refine_: candidates (ascending) in lds.cand_d / cand_s [count]; the selection lands in lds.kept_s / kept_d.
// Candidate c is kept iff no already-kept s has d(c, s) < d(c, query) (index.hpp:4040-4057).
template <int MT, int NCH, int R>
__device__ __forceinline__ int refine_candidates(const GraphView &gv, WaveLds &lds, int count, int needed,
                                                 WorkCounters &wc) {
	const int lane = lane_id();
	if (count < needed) {
		for (int i = lane; i < count; i += 64) {
			lds.kept_s[i] = lds.cand_s[i];
			lds.kept_d[i] = lds.cand_d[i];
		}
		wave_sync();
		return count;
	}
	if (lane == 0) {
		lds.kept_s[0] = lds.cand_s[0];
		lds.kept_d[0] = lds.cand_d[0];
	}
	wave_sync();
	int submitted = 1, consumed = 1;
	while (submitted < needed && consumed < count) {
		const uint32_t cs = lds.cand_s[consumed];
		const float cd = lds.cand_d[consumed];
		stage_row(lds.q2, gv.sp.vectors + (size_t)cs * gv.sp.V, gv.sp.V);
		const float c2 = MT == 1 ? wave_query_norm(gv.sp, lds.q2) : 0.f;
		// d(c, kept) in passes of R rows per lane group, nearest kept first; stop at the first pass that finds a kept
		// neighbour closer to c than the query is (index.hpp:4048-4051 breaks at the first such neighbour too)
		const int pass = R * (64 >> gv.sp.logG);
		bool bad = false;
		for (int off = 0; off < submitted && !bad; off += pass) {
			const int n = submitted - off < pass ? submitted - off : pass;
			wave_distances<MT, NCH, R>(gv.sp, lds.q2, c2, lds.kept_s + off, n, lds.dist);
			wc.distances += n;
			for (int o = 0; o < n; o += 64) {
				const bool b = (o + lane < n) && (lds.dist[o + lane] < cd);
				bad = bad || (__ballot(b) != 0ull);
			}
			wave_sync();
		}
		if (!bad) {
			if (lane == 0) {
				lds.kept_s[submitted] = cs;
				lds.kept_d[submitted] = cd;
			}
			submitted++;
			wave_sync();
		}
		consumed++;
	}
	return submitted;
}

// =========================================================================================================
// k_search — the search engine.  One persistent workgroup per compute unit: S walking waves, each taking one query at
// a time off a global counter (work stealing across the whole batch), and blockDim/64 - S scoring waves shared by all
// of them through the LDS mailboxes above.  While one walker is busy with its own bookkeeping (visited set, candidate
// list) the scoring waves load rows for the others, and a walker whose neighbours have finished gets every scoring
// wave of the CU — which is what shortens the tail of a batch (the slowest query) and the latency of a single query.
// =========================================================================================================
// One launch may carry several probe batches (vss_search_multi_device_begin): the queries of batch b are numbered
// b * batch_size .. and read / answered through the b-th entry of these tables; a plain probe is a launch of one batch.
constexpr int MAX_COALESCED = 32; // (round 5: 16 -> 32: the drain of a launch is paid once per launch, whatever it carries)
constexpr int PIPELINED_MAX_REGS = 4; // level_search_pipelined: candidate lists of at most this many registers (limit <= 256) ...
// ... in a 1024-thread workgroup (128 registers per lane).  Round 5: limits of 257-512 — the 8-register list — run as
// 768-thread workgroups (12 waves: 170 registers per lane), where the pipeline's state fits next to the list; four walkers
// and eight scoring waves, which is plenty for expansions that bring four to eight new rows.
constexpr int WIDE_LIST_THREADS = 768;
__host__ __device__ constexpr int pipelined_max_regs(int workgroup_threads) {
	return workgroup_threads <= WIDE_LIST_THREADS ? MAX_LIST_REGS : PIPELINED_MAX_REGS;
}
struct SearchArgs {
	GraphView gv;
	const float *queries[MAX_COALESCED]; // per batch: batch_size x q_stride floats
	uint32_t batch_size;
	uint32_t q_stride;
	uint32_t n_queries;   // queries to run in this launch (entries of `work` if given)
	uint32_t k, ef;
	uint32_t entry;
	int max_level;
	uint32_t tomb;        // index holds tombstones / a predicate is given: 1 = pending candidates in registers (RegQueue),
	                      // 2 = in the unbounded CandQueue (cand_buf)
	uint32_t hash_log2;   // visited set capacity
	uint32_t list_cap_max; // max(M, M0) rounded up to 64
	uint32_t walkers;     // S: walking waves per workgroup (the first S waves)
	uint32_t stage_cap;   // cells of the per-walker list-merge staging area in LDS (0 = none)
	uint32_t spec_active; // look one expansion ahead while at most this many walkers of the workgroup still run (0 = never)
	uint32_t touch_lines; // solo shape: bits 0-7 = 128-byte lines per row pulled into L2 one expansion ahead (RowTouch; 0 = off), TOUCH_LISTS = ListTouch
	                      // (the workgroup engine honours TOUCH_LISTS only)
	uint32_t crew;        // workgroup engine: bit 0 = the last walker of a workgroup runs its scoring waves as a crew (barriers, no
	                      // mailbox); bit 1 = the crew's scoring waves touch the neighbour lists of the rows they score (CrewTouch);
	                      // bit 2 = scoring waves on the walker's own SIMD take no rows (the pipelined walker computes while they
	                      // score and outranks them: they would finish last); bit 3 = no look-ahead list requests by a walker that
	                      // runs a crew (the lists are in L2 through the touches; the requests are ~100 instructions per expansion)
	uint32_t pipelined;   // workgroup engine: level_search_pipelined (host: no tombstones / predicate, register list, lists <= 64 cells)
	const uint32_t *work; // optional: list of query indices to run (retry pass), NULL = all
	uint32_t *queue;      // [queue_sel] next unclaimed position of the batch (zero at launch), [4..67] scrap.  Launches of a
	                      // context alternate between cells 0 and 2 and each zeroes the other one for its successor, so
	                      // that no memset has to travel down the stream ahead of every launch.
	uint32_t queue_sel;   // 0 or 2
	uint32_t *engine_error; // pinned host word (zero at launch): a walker gave up waiting
	uint32_t *drain_flag; // pinned host word, set to 1 when the LAST query of the launch has been handed out (may be NULL)
	uint32_t *done_count; // pinned host word (zero at launch; may be NULL): +1, released at system scope, per answered query —
	                      // the one-query probe's host thread waits on it instead of on the stream (results are in pinned memory)
	int64_t *out_keys[MAX_COALESCED];   // per batch: batch_size x k
	float *out_d[MAX_COALESCED];        // per batch: batch_size x k (may be NULL)
	uint32_t *out_count[MAX_COALESCED]; // per batch: batch_size
	uint32_t *out_stats;  // n_queries x 2 (may be NULL)
	uint32_t *status;     // n_queries: LEVEL_OK / LEVEL_OK_RETRIED / LEVEL_VISITED_OVERFLOW / LEVEL_QUEUE_OVERFLOW
	// A query that outgrows its LDS-resident visited set goes on over a table of 2^retry_log2 32-bit cells in HBM (one per walker).
	// Round 5 repeated such a query from the top over that table — instead of handing it back to the host for a second launch that
	// ran after everything else and lasted as long as its heaviest query (7 ms behind an 80 ms launch on the configs[4] shard,
	// profiles/r05_pmc_k_search_config4_shard_*.json); round 6 MOVES the set's contents there (the compact cells are invertible:
	// VisitedSet::migrate) and the query keeps what it has done.  nullptr = off (the host re-runs, as before; also what happens
	// to a query that outgrows this table too)
	uint32_t *retry_hash;
	uint32_t retry_log2;
	uint32_t *global_hash; // visited sets in HBM (grid x S x 2^hash_log2 words) or NULL = LDS
	float *list_buf;      // MemList storage (E == 0): grid x S x 2 x list_cap words
	uint32_t list_cap;
	float *cand_buf;      // CandQueue storage (tomb): grid x S x 2 x cand_cap words
	uint32_t cand_cap;
	uint32_t visited_compact; // workgroup engine, limits of 257-512: the FORM of the compact visited set laid over the 2^hash_log2
	                          // words of LDS (visited_compact.h: log2 of its 16-bit cells | key bits << 8; 0 = the plain 32-bit
	                          // set); host: slots < 2^25, first pass only
	unsigned long long *phase_ticks; // debug (VSS_PHASE_TIMERS): n_queries x VSS_PHASE_STRIDE
};

__host__ __device__ inline uint32_t align16(uint32_t x) {
	return (x + 15u) & ~15u;
}

// dynamic LDS bytes of one wave of the BUILD kernels (shared by host launch code and the kernels)
__host__ __device__ inline uint32_t wave_lds_bytes(uint32_t hash_log2, uint32_t V, uint32_t list_cap_max,
                                                   uint32_t cand_cap, bool hash_in_lds = true) {
	uint32_t b = 0;
	if (hash_in_lds)
		b += align16((1u << hash_log2) * 4);
	b += align16(V * 16) * 2;
	b += align16(list_cap_max * 4) * 2;
	b += align16(cand_cap * 4) * 2;
	b += align16((list_cap_max + 1) * 4) * 2;
	return b;
}

__device__ __forceinline__ void bind_visited(VisitedSet &v, uint32_t *table, uint32_t hash_log2) {
	v.table = table;
	v.mask = (1u << hash_log2) - 1;
	v.shift = 32 - hash_log2;
	v.limit = ((1u << hash_log2) / 8) * 7;
	v.count = 0;
	v.compact = 0;
}
// the compact form over the same bytes: 2^cells_log2 16-bit cells (VisitedSet, wave_primitives.h); filled to 3/4 at most
__device__ __forceinline__ void bind_visited_compact(VisitedSet &v, uint32_t form) { // (form: visited_compact.h — cells and key bits)
	const uint32_t cells_log2 = compact_visited::cells_log2_of(form);
	v.mask = (1u << cells_log2) - 1;
	v.limit = ((1u << cells_log2) / 4) * 3;
	v.compact = form;
}

// `global_hash` != nullptr: the visited set of this wave lives in HBM/L2 (large ef: a 64+ KiB table per wave would cut
// the occupancy to 1-2 waves per CU); otherwise it is carved from LDS.
__device__ __forceinline__ void carve_lds(WaveLds &lds, unsigned char *base, uint32_t hash_log2, uint32_t V,
                                          uint32_t list_cap_max, uint32_t cand_cap, uint32_t *global_hash = nullptr) {
	unsigned char *p = base;
	bind_visited(lds.visited, global_hash ? global_hash + ((size_t)blockIdx.x << hash_log2) : reinterpret_cast<uint32_t *>(p),
	             hash_log2);
	if (!global_hash)
		p += align16((1u << hash_log2) * 4);
	lds.q = reinterpret_cast<float4 *>(p);
	p += align16(V * 16);
	lds.q2 = reinterpret_cast<float4 *>(p);
	p += align16(V * 16);
	lds.ids = reinterpret_cast<uint32_t *>(p);
	p += align16(list_cap_max * 4);
	lds.dist = reinterpret_cast<float *>(p);
	p += align16(list_cap_max * 4);
	lds.cand_d = reinterpret_cast<float *>(p);
	p += align16(cand_cap * 4);
	lds.cand_s = reinterpret_cast<uint32_t *>(p);
	p += align16(cand_cap * 4);
	lds.kept_s = reinterpret_cast<uint32_t *>(p);
	p += align16((list_cap_max + 1) * 4);
	lds.kept_d = reinterpret_cast<float *>(p);
}

// LDS of the search engine: a header {exit flag, walkers still running}, S mailboxes, then per walker
// [visited set unless in HBM][staged query][ids][distances].
// stage_cap: cells of the list-merge staging area (>= the search limit for register lists, 0 = none)
__host__ __device__ inline uint32_t engine_slot_bytes(uint32_t hash_log2, uint32_t V, uint32_t list_cap_max, bool hash_in_lds,
                                                      uint32_t stage_cap) {
	return (hash_in_lds ? align16((1u << hash_log2) * 4) : 0) + align16(V * 16) + 4 * align16(list_cap_max * 4) +
	       2 * align16(stage_cap * 4);
}
__host__ __device__ inline uint32_t engine_lds_bytes(uint32_t walkers, uint32_t hash_log2, uint32_t V, uint32_t list_cap_max,
                                                     bool hash_in_lds, uint32_t stage_cap) {
	return ENGINE_HEADER_BYTES + walkers * engine_slot_bytes(hash_log2, V, list_cap_max, hash_in_lds, stage_cap);
}

struct EngineSlot {
	float4 *q;
	uint32_t *ids, *ids2; // job buffer 0 / 1
	float *dist, *dist2;
	uint32_t *hash; // LDS table, or nullptr when the visited sets live in HBM
	float *stage_d; // list-merge staging (nullptr if stage_cap == 0)
	uint32_t *stage_s;
};
__device__ __forceinline__ EngineSlot engine_slot(unsigned char *smem, uint32_t s, uint32_t hash_log2, uint32_t V,
                                                  uint32_t list_cap_max, bool hash_in_lds, uint32_t stage_cap) {
	unsigned char *p = smem + ENGINE_HEADER_BYTES + s * engine_slot_bytes(hash_log2, V, list_cap_max, hash_in_lds, stage_cap);
	EngineSlot e;
	e.hash = hash_in_lds ? reinterpret_cast<uint32_t *>(p) : nullptr;
	if (hash_in_lds)
		p += align16((1u << hash_log2) * 4);
	e.q = reinterpret_cast<float4 *>(p);
	p += align16(V * 16);
	e.ids = reinterpret_cast<uint32_t *>(p);
	p += align16(list_cap_max * 4);
	e.dist = reinterpret_cast<float *>(p);
	p += align16(list_cap_max * 4);
	e.ids2 = reinterpret_cast<uint32_t *>(p);
	p += align16(list_cap_max * 4);
	e.dist2 = reinterpret_cast<float *>(p);
	p += align16(list_cap_max * 4);
	e.stage_d = stage_cap ? reinterpret_cast<float *>(p) : nullptr;
	p += align16(stage_cap * 4);
	e.stage_s = stage_cap ? reinterpret_cast<uint32_t *>(p) : nullptr;
	return e;
}

// results of one query: the first `count` entries of the list, -1 / +inf beyond
template <int E>
__device__ __forceinline__ void emit_results(const GraphView &gv, int64_t *out_keys, float *out_d, int k, const WaveList<E> &L,
                                             int count) {
	const int lane = lane_id();
	for (int pos = count + lane; pos < k; pos += 64) { // the unused tail
		out_keys[pos] = -1ll;
		if (out_d)
			out_d[pos] = __builtin_inff();
	}
	const int base = lane * E - L.off; // list position of this lane's register 0 (blocked, right-aligned layout)
#pragma unroll
	for (int r = 0; r < E; ++r) {
		const int pos = base + r;
		if (pos >= 0 && pos < count) {
			out_keys[pos] = gv.keys[L.s[r] & ~EXPANDED_BIT];
			if (out_d)
				out_d[pos] = L.d[r];
		}
	}
}
__device__ __forceinline__ void emit_results(const GraphView &gv, int64_t *out_keys, float *out_d, int k, const MemList &L,
                                             int count) {
	for (int pos = lane_id(); pos < k; pos += 64) {
		const bool valid = pos < count;
		out_keys[pos] = valid ? gv.keys[L.s[pos] & ~EXPANDED_BIT] : -1ll;
		if (out_d)
			out_d[pos] = valid ? L.d[pos] : __builtin_inff();
	}
}

// A scoring wave in crew mode (see CrewBox): parked at the first barrier until the walker offers rows, scores its share,
// meets everybody at the second barrier.  h = this wave's number among the H scoring waves.
//
// ListTouch by the crew: the walker's next candidate is, more often than not, one of the rows being scored right now, and
// the moment the scores arrive the walker asks for that row's neighbour list — a dependent HBM round trip on the critical
// path.  Each scoring wave therefore pulls the lines of the neighbour lists of ITS rows into L2 (one dword per 128-byte
// line; the values are never used) right after it has issued its row loads: the list is an L2 hit when the walker wants
// it.  Costs 128-256 bytes per scored row next to the 3 KiB of the row itself; crews only run latency-bound work.
struct CrewTouch {
	static constexpr bool lds_only_sync = true;
	const GraphView *gv;
	const uint32_t *ids; // this wave's share
	int rows;
	uint32_t lines;      // 128-byte lines per list of this level (1 or 2), 0 = off
	uint32_t level;      // the graph level the rows were gathered on: their lists of THAT level are the ones read next
	uint32_t *sink;
#ifdef VSS_PHASE_TIMERS
	unsigned long long *t_issue; // (profiling builds: when the first scoring wave has issued its row loads)
#endif
	__device__ __forceinline__ void operator()() const {
#ifdef VSS_PHASE_TIMERS
		*t_issue = __builtin_readcyclecounter();
#endif
		if (lines) {
			const uint32_t l = (uint32_t)lane_id();
			const uint32_t row = lines == 2 ? l >> 1 : l, line = lines == 2 ? l & 1u : 0u;
			if ((int)row < rows) {
				// (an upper level: the list's place comes from upper_off[] — a dependent load, in the shadow of this wave's own
				//  row loads; the walker's read of upper_off[] for the row it moves to is an L2 hit as well afterwards)
				const uint32_t *list = level == 0 ? gv->links0 + (size_t)ids[row] * gv->M0
				                                  : gv->links_up + ((size_t)gv->upper_off[ids[row]] + (level - 1)) * gv->M;
				*sink = list[line * 32u];
			}
		}
	}
};
template <int MT, int NCH, int R>
__device__ __forceinline__ void crew_help(unsigned char *smem, const SearchArgs &a, const CrewBox *crew, int wave, int S,
                                          int waves, bool hash_in_lds) {
	const int RG = 64 >> a.gv.sp.logG;
	const uint32_t touch_on = (a.crew & CREW_TOUCH) ? 1u : 0u; // (lists of at most 64 cells — one or two 128-byte lines: host)
	uint32_t sink = 0;
	// set once the walker is known (first job): this wave's number h among the H scoring waves that take rows, the reciprocal
	// of H (shares without an integer division per expansion: n / H by 16-bit reciprocal, exact while n * H < 2^16), and the
	// walker's two job buffers
	int h = 0, H = 0;
	uint32_t inv_h = 0, slot = ~0u;
	const float4 *q = nullptr;
	const uint32_t *ids0 = nullptr, *ids1 = nullptr;
	float *dist0 = nullptr, *dist1 = nullptr;
#ifdef VSS_PHASE_TIMERS
	unsigned long long *acc = reinterpret_cast<unsigned long long *>(smem + ENGINE_SCRAP_OFFSET); // free in crew mode
#endif
	for (;;) {
		lds_barrier();
		VSS_TICK(th0);
		// the whole box in ONE LDS round trip (it is 16 bytes, 16-byte aligned): a scoring wave's prologue is a chain of
		// dependent LDS reads — box, ids, then the row addresses — and four waves per SIMD issue theirs side by side
		const lds_u32x4 w = *VSS_LDS_PTR(const lds_u32x4_as3, crew);
		const int n = uniform((int)w.x);
		if (n < 0)
			break;
		const float qa2 = __uint_as_float((uint32_t)uniform((int)w.y));
		const uint32_t where_level = (uint32_t)uniform((int)w.z);
		const uint32_t where = where_level & 0xFFu, level = where_level >> 8;
		if ((where >> 1) != slot) { // (once: a crew serves one walker until the launch is over)
			slot = where >> 1;
			const EngineSlot es = engine_slot(smem, slot, a.hash_log2, a.gv.sp.V, a.list_cap_max, hash_in_lds, a.stage_cap);
			q = es.q, ids0 = es.ids, ids1 = es.ids2, dist0 = es.dist, dist1 = es.dist2;
			// Which scoring waves take rows: all of them — or, when the walker works through its accept phase while they score
			// (CREW_SPARE_SIMD), only those on the other SIMDs: at priority 2 the walker keeps its own SIMD to itself, the
			// scoring waves next to it would be the last to finish, and everybody meets at the second barrier.
			const unsigned char *simd_of = smem + ENGINE_SIMD_OFFSET;
			const bool spare = (a.crew & CREW_SPARE_SIMD) != 0;
			const unsigned char wsimd = simd_of[slot];
			int others = 0;
			for (int w = S; w < waves; ++w)
				others += simd_of[w] != wsimd;
			const bool use_spare = spare && others >= 4;
			H = 0, h = -1;
			for (int w = S; w < waves; ++w) {
				const bool takes = !use_spare || simd_of[w] != wsimd;
				if (w == wave)
					h = takes ? H : -1;
				H += takes;
			}
			H = uniform(H), h = uniform(h);
			inv_h = (65536u + (uint32_t)H - 1u) / (uint32_t)H;
		}
		const uint32_t *ids = (where & 1u) ? ids1 : ids0;
		float *dist = (where & 1u) ? dist1 : dist0;
		const uint32_t un = (uint32_t)n;
		const int ceil_nh = (un + (uint32_t)H - 1u) * (uint32_t)H < 65536u ? (int)(((un + (uint32_t)H - 1u) * inv_h) >> 16)
		                                                                    : (int)((un + (uint32_t)H - 1u) / (uint32_t)H);
		const int per = (ceil_nh + RG - 1) & ~(RG - 1); // a multiple of the rows a register slot handles side by side
		const int lo = h < 0 ? n : (h * per < n ? h * per : n);
		const int hi = lo + per < n ? lo + per : n;
#ifdef VSS_PHASE_TIMERS
		unsigned long long t_issue = th0;
#endif
		if (hi > lo) {
			asm volatile("" ::"v"(sink)); // the previous expansion's touches: long landed
			CrewTouch hook;
			hook.gv = &a.gv, hook.ids = ids + lo, hook.rows = hi - lo, hook.level = level, hook.sink = &sink;
			hook.lines = touch_on ? ((level == 0 ? a.gv.M0 : a.gv.M) > 32 ? 2u : 1u) : 0u;
#ifdef VSS_PHASE_TIMERS
			hook.t_issue = &t_issue;
#endif
			// a share of one or two register slots takes the narrow variants (the wide one would load clamped duplicates of
			// its last row); every variant reduces a row with the same lanes in the same order: same bits
			const int slots = per >> (6 - (int)a.gv.sp.logG);
			if (slots <= 1)
				wave_distances<MT, NCH, 1>(a.gv.sp, q, qa2, ids + lo, hi - lo, dist + lo, hook);
			else if (slots == 2)
				wave_distances<MT, NCH, 2>(a.gv.sp, q, qa2, ids + lo, hi - lo, dist + lo, hook);
			else
				wave_distances<MT, NCH, R>(a.gv.sp, q, qa2, ids + lo, hi - lo, dist + lo, hook);
		}
		VSS_TICK(th2);
		lds_barrier();
#ifdef VSS_PHASE_TIMERS
		if (h == 0 && lane_id() == 0) { // the first scoring wave's view: prologue, rows + arithmetic, wait at the second barrier
			acc[0] += t_issue - th0;
			acc[1] += th2 - t_issue;
			acc[2] += __builtin_readcyclecounter() - th2;
		}
#endif
	}
	asm volatile("" ::"v"(sink));
}

// E = registers of the candidate list (2, 4, 8), or 0 = MemList in HBM for limits beyond 64 * MAX_LIST_REGS
// (Round 4 measured a 12-wave variant for 1536-dimensional rows — __launch_bounds__(768), 170 registers: 4 rows in flight per
// scoring wave and the pipelined level search next to an 8-register list — at exactly the 16-wave kernel's rate while the visited
// sets of these limits still lived in HBM (profiles/r04f_wide_rows_1536_workgroup_shapes.txt) and dropped it.  Round 5, with the
// compact sets in LDS, brought it back for the 8-register list at every row width: profiles/r05c_*, r05j_*, r05m_*.)
// THREADS = the largest workgroup the instantiation is launched with: 1024 (16 waves, 128 registers per lane), or
// WIDE_LIST_THREADS for the pipelined 8-register list (round 5)
template <int MT, int NCH, int R, int E, int THREADS = 1024>
__global__ __launch_bounds__(THREADS) void k_search(SearchArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = lane_id();
	const uint32_t wave = (uint32_t)uniform((int)(threadIdx.x >> 6));
	const uint32_t S = a.walkers;
	const bool hash_in_lds = a.global_hash == nullptr;
	uint32_t *exit_flag = reinterpret_cast<uint32_t *>(smem);
	uint32_t *walkers_left = exit_flag + 1;
	CrewBox *crew = reinterpret_cast<CrewBox *>(smem + ENGINE_CREW_OFFSET);
	Mailbox *boxes = reinterpret_cast<Mailbox *>(smem + ENGINE_BOX_OFFSET);
	unsigned long long *scrap = reinterpret_cast<unsigned long long *>(smem + ENGINE_SCRAP_OFFSET);
	if (threadIdx.x == 0) {
		*exit_flag = 0;
		*walkers_left = S;
		crew->n = 0, crew->qa2 = 0.f, crew->walker = 0, crew->on = 0;
		a.queue[a.queue_sel ^ 2u] = 0; // the next launch's counter (nobody uses it during this one)
	}
	VSS_TRACE(a.gv.sp, 30, blockDim.x);
	VSS_TRACE(a.gv.sp, 31, S);
	if (lane == 0) // which SIMD this wave landed on (HW_ID bits 5:4): a crew whose walker computes while it scores spares that SIMD
		smem[ENGINE_SIMD_OFFSET + wave] = (unsigned char)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
	if (threadIdx.x < ENGINE_BOXES) {
		boxes[threadIdx.x].ticket = 0;
		boxes[threadIdx.x].done = 0;
		boxes[threadIdx.x].qa2_bits = 0;
		boxes[threadIdx.x].slots = R;
	}
	__syncthreads();

	if (wave >= S) { // ---------------------------------------------------------------- scoring waves
		for (;;) {
			bool worked = false;
			VSS_TRACE_INC(a.gv.sp, 24);
			for (uint32_t s = 0; s < 2 * S; ++s) { // mailbox s = job buffer (s & 1) of walker (s >> 1)
				const unsigned long long t = VSS_LDS_LOAD(lds_u64, &boxes[s].ticket);
				if (uniform((int)((uint32_t)t < (uint32_t)(t >> 32)))) {
					const EngineSlot es = engine_slot(smem, s >> 1, a.hash_log2, a.gv.sp.V, a.list_cap_max, hash_in_lds, a.stage_cap);
					worked |= pool_score<MT, NCH, R>(&boxes[s], scrap, a.gv.sp, es.q, (s & 1) ? es.ids2 : es.ids,
					                                 (s & 1) ? es.dist2 : es.dist);
				}
			}
			if (uniform((int)VSS_LDS_LOAD(lds_u32, exit_flag)))
				return;
			if (uniform((int)VSS_LDS_LOAD_ACQ(lds_u32, &crew->on))) { // one walker left: its crew, behind barriers, until it is done
				crew_help<MT, NCH, R>(smem, a, crew, (int)wave, (int)S, (int)(blockDim.x >> 6), hash_in_lds);
				return;
			}
			if (!worked)
				__builtin_amdgcn_s_sleep(VSS_SCORER_IDLE_SLEEP);
		}
	}

	// -------------------------------------------------------------------------------- walking waves
	__builtin_amdgcn_s_setprio(2); // the serial bookkeeping of a walker is the critical path of its query
	const EngineSlot es = engine_slot(smem, wave, a.hash_log2, a.gv.sp.V, a.list_cap_max, hash_in_lds, a.stage_cap);
	const size_t gslot = (size_t)blockIdx.x * S + wave; // this walker's scratch in HBM
	WaveLds lds;
	bind_visited(lds.visited, hash_in_lds ? es.hash : a.global_hash + (gslot << a.hash_log2), a.hash_log2);
	if constexpr (E == MAX_LIST_REGS) { // (limits of 257-512: the only instantiation that carries the compact form's code)
		if (a.visited_compact)
			bind_visited_compact(lds.visited, a.visited_compact);
	}
	lds.q = es.q, lds.ids = es.ids, lds.dist = es.dist;
	lds.q2 = nullptr, lds.kept_s = nullptr, lds.kept_d = nullptr;
	lds.cand_d = nullptr, lds.cand_s = nullptr; // (rounds 2-5: staging of the batched list merge; the slot's few words are the spill box now)
	lds.touch_lines = a.touch_lines & TOUCH_LISTS;   // latency-bound launches (host): ListTouch from the first expansion on
	PoolScorer<MT, NCH, R> score {&boxes[2 * wave], exit_flag, a.engine_error, walkers_left, (blockDim.x >> 6) - S, crew, wave,
	                              ((a.crew & CREW_ON) && !a.spec_active) ? 1u : 0u, (a.crew & CREW_NO_REQUESTS) ? 1u : 0u,
	                              (a.crew & CREW_TOUCH) ? 1u : 0u};
	const SpecBuffers sb {es.ids, es.ids2, es.dist, es.dist2};
	CandQueue cq;
	cq.bind(a.cand_buf + gslot * 2 * a.cand_cap, reinterpret_cast<uint32_t *>(a.cand_buf + gslot * 2 * a.cand_cap) + a.cand_cap,
	        (int)a.cand_cap);
	typename std::conditional<E == 0, MemList, WaveList<(E == 0 ? 1 : E)>>::type L;
	if constexpr (E == 0)
		L.bind(a.list_buf + gslot * 2 * a.list_cap, reinterpret_cast<uint32_t *>(a.list_buf + gslot * 2 * a.list_cap) + a.list_cap);
	const int limit = a.ef > a.k ? a.ef : a.k; // expansion = max(ef, wanted), index.hpp:2908

	// (the 8-register list's instantiations only — limits of 257-512, where the compact set's overflows are a per-cent matter:
	//  every other instantiation stays byte-identical to round 4's, registers included)
	constexpr bool CAN_MOVE = E == MAX_LIST_REGS;
	if constexpr (CAN_MOVE) {
		if (a.retry_hash && hash_in_lds && es.stage_d) { // where this walker's visited set moves when it outgrows LDS
			lds.spill_box = reinterpret_cast<uint32_t *>(es.stage_d);
			if (lane == 0) {
				const uintptr_t t = (uintptr_t)(a.retry_hash + (gslot << a.retry_log2));
				lds.spill_box[0] = (uint32_t)t, lds.spill_box[1] = (uint32_t)(t >> 32), lds.spill_box[2] = a.retry_log2, lds.spill_box[3] = 0;
			}
			lds_sync();
		}
	}
	for (;;) {
		// every lane executes the atomic, lane 0 on the queue head and lane i on scrap word i (no lane-0 branch, see pool_score)
		const uint32_t idx = (uint32_t)uniform((int)atomicAdd(lane == 0 ? a.queue + a.queue_sel : a.queue + 4 + lane, 1u));
		if (idx >= a.n_queries)
			break;
		// the queue is dry from here on: compute units start to fall idle, the host may issue the next launch
		if (idx + 1 == a.n_queries && a.drain_flag)
			__hip_atomic_store(a.drain_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		const uint32_t qi = a.work ? a.work[idx] : idx;
		VSS_TRACE(a.gv.sp, 19, 1u);
		// which batch of the launch, and which of its queries (the tables are read with wave-uniform indices: scalar loads
		// from the kernel arguments)
		const uint32_t batch = qi / a.batch_size, row = qi - batch * a.batch_size;
		stage_query(lds.q, a.queries[batch] + (size_t)row * a.q_stride, a.gv.dim, a.gv.sp.V);
		VSS_TRACE(a.gv.sp, 19, 2u);
		const float qa2 = MT == 1 ? wave_query_norm(a.gv.sp, lds.q) : 0.f;
		VSS_LDS_STORE(lds_u32, &boxes[2 * wave].qa2_bits, __float_as_uint(qa2));
		VSS_LDS_STORE(lds_u32, &boxes[2 * wave + 1].qa2_bits, __float_as_uint(qa2));
		lds.ids = es.ids, lds.dist = es.dist;
		WorkCounters wc = {};
		VSS_TICK(tq0);
		// (the look-ahead of the descent: not next to the 8-register list — its kernels have no register to spare, and a descent
		//  is 1-2 % of a query at those limits)
		uint32_t closest = descend<MT, (E >= 1 && E <= 4)>(a.gv, lds, qa2, a.entry, a.max_level, 0, score, wc);
		VSS_TICK(tq1);
		VSS_ACC(t_descend, tq0, tq1);
		VSS_TRACE(a.gv.sp, 19, 3u);
		int rc;
		// neighbour lists in flight: the 1024-thread workgroup allows 128 registers per lane — one list only where the row
		// window (dimension 1536) or the candidate list (8 registers) already fills them
		// (round 6 measured two lists in flight next to the blocked 8-register list in its 12-wave instantiation, which has the
		//  registers for it now: 262k -> 251k queries/s at 10M x 768, ef 512 — the second request and its bookkeeping cost the accept
		//  phase more than the extra hits save the gather: profiles/r06c_wide_lists_phase_ticks_10m768_prof.txt.  One list stays.)
		constexpr int PK = (NCH == 6 || NCH == 4 || E == 0 || E >= 8) ? 1 : 2;
		if (a.tomb == 1) { // few rejected rows expected: the pending candidates stay in registers (host: limits within the register lists only)
			if constexpr (E == 2 || E == 4 || E == 8) {
				// twice the result list (the queue fills up while the result list is still filling); as long as the list itself
				// where that is all the registers there are
				RegQueue<(E == 8 ? E : 2 * E)> rq;
				rc = level_search_impl<MT, false, true, 1>(a.gv, lds, qa2, closest, EMPTY_SLOT, 0, limit, L, rq, score, wc);
			} else {
				rc = LEVEL_QUEUE_OVERFLOW;
			}
		} else if (a.tomb)
			rc = level_search_impl<MT, false, true, 1>(a.gv, lds, qa2, closest, EMPTY_SLOT, 0, limit, L, cq, score, wc);
		else if (a.spec_active)
			rc = level_search_spec<MT>(a.gv, lds, sb, qa2, closest, limit, L, score, a.spec_active, wc);
		else if (E > 0 && E <= pipelined_max_regs(THREADS) && a.pipelined) {
			// accept phase in the shadow of the successor's row loads (host: lists of at most 64 cells).  Limits beyond 256 — an
			// 8-register list — keep the plain order in a 1024-thread workgroup: the pipeline's state next to it does not fit its
			// 128 registers (112 bytes of scratch per lane measured); the 768-thread instantiation has room.
			if constexpr (E > 0 && E <= pipelined_max_regs(THREADS))
				rc = level_search_pipelined<MT, PK>(a.gv, lds, sb, qa2, closest, limit, L, score, wc);
			else
				rc = LEVEL_INTERNAL;
		} else
			rc = level_search_impl<MT, false, false, PK>(a.gv, lds, qa2, closest, EMPTY_SLOT, 0, limit, L, cq, score, wc);
		// (a compact set that outgrew LDS has MOVED to this walker's table in HBM meanwhile — gather_neighbors — and the query went
		//  on; LEVEL_VISITED_OVERFLOW from here means that table overflowed too: the host re-runs such a query with a larger one)
		bool moved = false;
		if constexpr (CAN_MOVE) {
			if (lds.spill_box) {
				moved = uniform((int)lds.spill_box[3]) != 0;
				if (moved) { // the next query starts in LDS again (opaque copies: no address arithmetic hoisted into scalar registers)
					uint32_t hl = a.hash_log2;
					asm volatile("" : "+s"(hl));
					bind_visited(lds.visited, es.hash, hl);
					if (a.visited_compact)
						bind_visited_compact(lds.visited, a.visited_compact);
					if (lane == 0)
						lds.spill_box[3] = 0;
					lds_sync();
				}
			}
		}
		VSS_TRACE(a.gv.sp, 19, 4u);
		const int count = rc == LEVEL_OK ? (L.size < (int)a.k ? L.size : (int)a.k) : 0;
		emit_results(a.gv, a.out_keys[batch] + (size_t)row * a.k, a.out_d[batch] ? a.out_d[batch] + (size_t)row * a.k : nullptr,
		             (int)a.k, L, count);
		if (lane == 0) {
			a.out_count[batch][row] = count;
			a.status[qi] = (rc == LEVEL_OK && moved) ? (uint32_t)LEVEL_OK_RETRIED : (uint32_t)rc;
			if (a.out_stats) {
				a.out_stats[2 * qi] = wc.distances;
				a.out_stats[2 * qi + 1] = wc.cycles;
			}
#ifdef VSS_PHASE_TIMERS
			if (a.phase_ticks) {
				unsigned long long *o = a.phase_ticks + VSS_PHASE_STRIDE * (size_t)qi;
				o[0] = wc.t_pick, o[1] = wc.t_gather, o[2] = wc.t_dist, o[3] = wc.t_accept, o[4] = wc.t_descend;
				o[5] = __builtin_readcyclecounter() - tq0;
				o[6] = wc.t_sync1, o[7] = wc.t_look, o[8] = wc.t_slice, o[9] = wc.t_sync2, o[10] = wc.t_team_passes;
				o[11] = wc.t_solo_passes;
				if (score.crew_on) { // the first scoring wave's accumulators (crew_help): prologue, rows + arithmetic, second barrier
					unsigned long long *acc = reinterpret_cast<unsigned long long *>(smem + ENGINE_SCRAP_OFFSET);
					o[6] = acc[0], o[8] = acc[1], o[10] = acc[2];
					acc[0] = acc[1] = acc[2] = 0;
				}
			}
#endif
		}
		if (a.done_count) {
			__threadfence_system(); // every lane's result cells, then the count
			if (lane == 0)
				__hip_atomic_fetch_add(a.done_count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
	VSS_TRACE(a.gv.sp, 19, 5u);
	if (score.crew_on) { // the last walker of the workgroup, its scoring waves parked at the crew's barrier: send them home
		score.dismiss_crew();
		return;
	}
	if (lane == 0 && VSS_LDS_ADD(lds_u32, walkers_left, 0xFFFFFFFFu) == 1u)
		VSS_LDS_STORE(lds_u32, exit_flag, 1u);
	VSS_TRACE(a.gv.sp, 19, 6u);
}

// =========================================================================================================
// k_search_solo — the search engine's shape for FEW queries (the single-query probe of HNSW_INDEX_SCAN, reference
// hnsw_index_scan.cpp:43-90 / hnsw_index.cpp:315-341, and small batches): one 64-thread workgroup = one wave per query,
// and the walking wave scores its own rows.  No mailbox, no second wave to wake: an expansion is the candidate's list
// (requested one expansion ahead, ListPrefetch), ONE round of row loads with every row of the expansion in flight at once
// (R register slots x 64 / G rows: all 32 rows of a level-0 list at dimension 128 are 16 float4 per lane), the reduction,
// and the accept phase.  Without a 1024-thread workgroup the 128-register ceiling of k_search is gone, which is what
// makes the wide row window possible.  Rows are reduced by the same lanes in the same order as everywhere else
// (wave_distances), so ids, distance bits and work counters are those of k_search and of the oracle.
// Work is handed out through the same global counter (a launch may carry more queries than waves).
// =========================================================================================================
// (amdgpu_waves_per_eu(1, 2): LDS admits a handful of these waves per compute unit anyway; telling the compiler so keeps its
// scheduler from trading the row window's registers for an occupancy nobody can use)
// T > 1: a team — T - 1 helper waves score a share of every expansion's rows (TeamScorer above); R is then the row window
// of EACH wave.
template <int MT, int NCH, int R, int E, int T = 1>
__global__ __launch_bounds__(64 * T) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_search_solo(SearchArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	TeamBox &team_box = *reinterpret_cast<TeamBox *>(smem); // teams: the first bytes of the workgroup's LDS (host: + TEAM_BOX_BYTES)
	const int lane = lane_id();
	WaveLds lds;
	carve_lds(lds, smem + (T > 1 ? TEAM_BOX_BYTES : 0), a.hash_log2, a.gv.sp.V, a.list_cap_max, a.stage_cap, a.global_hash);
	if (!a.stage_cap)
		lds.cand_d = nullptr, lds.cand_s = nullptr; // no staging area: the list merges one by one
	lds.touch_lines = a.touch_lines;
	if constexpr (T > 1) {
		if (threadIdx.x >= 64) {
			team_help<MT, NCH, R, T>(lds, a.gv.sp, &team_box, (int)(threadIdx.x >> 6), a.touch_lines & 0xFFu);
			return;
		}
	}
	if (blockIdx.x == 0 && lane == 0)
		a.queue[a.queue_sel ^ 2u] = 0; // the next launch's counter (nobody uses it during this one)
	const size_t gslot = blockIdx.x; // this walker's scratch in HBM
	typename std::conditional<T == 1, SoloScorer<MT, NCH, R, true>, TeamScorer<MT, NCH, R, T>>::type score;
	if constexpr (T > 1)
		score.box = &team_box;
	CandQueue cq;
	cq.bind(a.cand_buf + gslot * 2 * a.cand_cap, reinterpret_cast<uint32_t *>(a.cand_buf + gslot * 2 * a.cand_cap) + a.cand_cap,
	        (int)a.cand_cap);
	typename std::conditional<E == 0, MemList, WaveList<(E == 0 ? 1 : E)>>::type L;
	if constexpr (E == 0)
		L.bind(a.list_buf + gslot * 2 * a.list_cap, reinterpret_cast<uint32_t *>(a.list_buf + gslot * 2 * a.list_cap) + a.list_cap);
	const int limit = a.ef > a.k ? a.ef : a.k; // expansion = max(ef, wanted), index.hpp:2908
	for (;;) {
		// every lane executes the atomic, lane 0 on the queue head and lane i on scrap word i (no lane-0 branch, see pool_score)
		const uint32_t idx = (uint32_t)uniform((int)atomicAdd(lane == 0 ? a.queue + a.queue_sel : a.queue + 4 + lane, 1u));
		if (idx >= a.n_queries)
			break;
		if (idx + 1 == a.n_queries && a.drain_flag)
			__hip_atomic_store(a.drain_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		const uint32_t qi = a.work ? a.work[idx] : idx;
		const uint32_t batch = qi / a.batch_size, row = qi - batch * a.batch_size;
		stage_query(lds.q, a.queries[batch] + (size_t)row * a.q_stride, a.gv.dim, a.gv.sp.V);
		const float qa2 = MT == 1 ? wave_query_norm(a.gv.sp, lds.q) : 0.f;
		WorkCounters wc = {};
		VSS_TICK(tq0);
		const uint32_t closest = descend<MT, (E >= 1 && E <= 4)>(a.gv, lds, qa2, a.entry, a.max_level, 0, score, wc);
		VSS_TICK(tq1);
		VSS_ACC(t_descend, tq0, tq1);
		int rc;
		if (a.tomb == 1) { // few rejected rows expected: the pending candidates stay in registers (host: limits within the register lists only)
			if constexpr (E == 1 || E == 2 || E == 4 || E == 8) {
				RegQueue<2 * E> rq;
				rc = level_search_impl<MT, false, true>(a.gv, lds, qa2, closest, EMPTY_SLOT, 0, limit, L, rq, score, wc);
			} else {
				rc = LEVEL_QUEUE_OVERFLOW;
			}
		} else if (a.tomb)
			rc = level_search_impl<MT, false, true>(a.gv, lds, qa2, closest, EMPTY_SLOT, 0, limit, L, cq, score, wc);
		else
			// (round 6 measured the team running level_search_pipelined — the helpers score, the walker accepts meanwhile — and dropped
			//  it: 222 against 196 us per query at 1M x 128; what the hidden accept phase saves, the pipeline's heavier pick and its
			//  second pair of barriers cost again: profiles/r06d_solo_phase_1m128_team_pipelined_not_kept.txt)
			rc = level_search_impl<MT, false, false>(a.gv, lds, qa2, closest, EMPTY_SLOT, 0, limit, L, cq, score, wc);
		const int count = rc == LEVEL_OK ? (L.size < (int)a.k ? L.size : (int)a.k) : 0;
		emit_results(a.gv, a.out_keys[batch] + (size_t)row * a.k, a.out_d[batch] ? a.out_d[batch] + (size_t)row * a.k : nullptr,
		             (int)a.k, L, count);
		if (lane == 0) {
			a.out_count[batch][row] = count;
			a.status[qi] = (uint32_t)rc;
			if (a.out_stats) {
				a.out_stats[2 * qi] = wc.distances;
				a.out_stats[2 * qi + 1] = wc.cycles;
			}
#ifdef VSS_PHASE_TIMERS
			if (a.phase_ticks) {
				unsigned long long *o = a.phase_ticks + VSS_PHASE_STRIDE * (size_t)qi;
				o[0] = wc.t_pick, o[1] = wc.t_gather, o[2] = wc.t_dist, o[3] = wc.t_accept, o[4] = wc.t_descend;
				o[5] = __builtin_readcyclecounter() - tq0;
				o[6] = wc.t_sync1, o[7] = wc.t_look, o[8] = wc.t_slice, o[9] = wc.t_sync2, o[10] = wc.t_team_passes;
				o[11] = wc.t_solo_passes;
			}
#endif
		}
		if (a.done_count) {
			__threadfence_system(); // every lane's result cells, then the count
			if (lane == 0)
				__hip_atomic_fetch_add(a.done_count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		wave_sync(); // the results are on their way before the wave's buffers are reused
	}
	if constexpr (T > 1) { // dismiss the helpers
		if (lane == 0)
			team_box.n = -1;
		__syncthreads();
	}
}

// =========================================================================================================
// Bulk build, phase A — one wave per new node: descent + per-level search + refine_, writes the node's own lists
// and emits one reverse-link request per selected neighbour.  The graph is read-only during this phase.
// =========================================================================================================
// Bit 31 of a request's source marks a re-linked (reused) node: its stored vector is still the OLD one while the links
// are repaired, so a distance cached from its request (taken with the NEW vector) must not stand in for the
// successor distance the reference recomputes from storage (index.hpp:3706-3712).
constexpr uint32_t REUSED_SOURCE = 0x80000000u;

struct BuildArgs {
	GraphView gv;
	uint32_t first_slot;  // nodes first_slot .. first_slot + n_nodes - 1 form the batch
	uint32_t n_nodes;
	const uint8_t *levels; // per slot
	uint32_t entry;
	int max_level;
	uint32_t top_limit;   // limit of the insert search = ef_construction (config.expansion, index.hpp:3648)
	uint32_t hash_log2;
	uint32_t list_cap_max;
	// reverse-link requests (SoA) + counters
	uint32_t *req_list;   // target list id
	uint32_t *req_src;    // the new node
	float *req_d;         // d(new, target)
	uint32_t *counters;   // [0] = number of requests, [1] = touched lists, [2] = scatter cursor, [3] = error flag
	uint32_t req_capacity;
	uint32_t *global_hash; // visited sets in HBM (grid x 2^hash_log2 words) or NULL = LDS
	const uint32_t *work;  // optional: indices (within the batch) of the nodes to run (retry pass), NULL = all
	uint32_t *node_status; // per node of the batch: 0 done, 1 visited-set overflow (node must be re-run)
	uint32_t node_req_cap; // requests one node can emit: M * (highest level in the batch + 1)
	unsigned long long *work_stats; // [0] += distances computed, [1] += nodes expanded (roofline accounting)
	// Rows that take over a tombstoned slot (usearch update(), index.hpp:2801-2859).  All NULL for plain appends.
	const uint32_t *row_slot; // per batch node: its slot (NULL: first_slot + node)
	const uint32_t *row_src;  // per batch node: row of `pending` holding its NEW vector, EMPTY_SLOT = vector already in place
	const float4 *pending;
	uint32_t *parked;         // per batch node (level_hi + 1) x list_cap_max words: the new lists of reused nodes, which stay
	uint32_t parked_stride;   //   reachable through stale links and so must look blank until the whole batch has searched
	float *list_buf;          // MemList storage (ef_construction > 64 * MAX_LIST_REGS): grid x 2 x list_cap words
	uint32_t list_cap;
	uint32_t cand_lds_cap;    // LDS cells of the dumped candidate list: top_limit, or a token 16 when it stays in HBM
};

// Occupancy target: four waves per SIMD wherever the row window allows it (the build is bound by the rows in flight per
// compute unit; left alone the register allocator lands a few registers above the 128 that four waves permit).
#ifndef VSS_BUILD_WAVES_PER_EU
#define VSS_BUILD_WAVES_PER_EU(NCH, E) ((NCH) == 6 ? 2 : (((E) >= 8 || (NCH) == 4) ? 3 : 4))
#endif
template <int MT, int NCH, int R, int E>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(VSS_BUILD_WAVES_PER_EU(NCH, E), 8))) void
k_build_phase_a(BuildArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = lane_id();
	const uint32_t node = a.work ? a.work[blockIdx.x] : blockIdx.x;
	const uint32_t slot = a.row_slot ? a.row_slot[node] : a.first_slot + node;
	const uint32_t src = a.row_src ? a.row_src[node] : EMPTY_SLOT;
	WaveLds lds;
	carve_lds(lds, smem, a.hash_log2, a.gv.sp.V, a.list_cap_max, a.cand_lds_cap, a.global_hash);
	// the node's reverse-link requests are buffered in LDS and published only when every level succeeded, so a node
	// that overflows its visited set can simply be re-run
	uint32_t *req_l = reinterpret_cast<uint32_t *>(
	    smem + wave_lds_bytes(a.hash_log2, a.gv.sp.V, a.list_cap_max, a.cand_lds_cap, a.global_hash == nullptr));
	float *req_dd = reinterpret_cast<float *>(req_l + a.node_req_cap);
	stage_row(lds.q, src == EMPTY_SLOT ? a.gv.sp.vectors + (size_t)slot * a.gv.sp.V : a.pending + (size_t)src * a.gv.sp.V,
	          a.gv.sp.V);
	const float qa2 = MT == 1 ? wave_query_norm(a.gv.sp, lds.q) : 0.f;
	WorkCounters wc = {};
	const int target = a.levels[slot];
	const SoloScorer<MT, NCH, R> score;
	uint32_t closest = descend<MT>(a.gv, lds, qa2, a.entry, a.max_level, target, score, wc);
	typename std::conditional<E == 0, MemList, WaveList<(E == 0 ? 1 : E)>>::type L;
	if constexpr (E == 0) // ef_construction beyond the register lists: the candidate list lives in HBM
		L.bind(a.list_buf + (size_t)blockIdx.x * 2 * a.list_cap,
		       reinterpret_cast<uint32_t *>(a.list_buf + (size_t)blockIdx.x * 2 * a.list_cap) + a.list_cap);
	CandQueue unused_queue;
	uint32_t n_req = 0;
	for (int level = target < a.max_level ? target : a.max_level; level >= 0; --level) {
		// (one neighbour list in flight: a second one costs the registers that decide between 3 and 4 waves per SIMD)
		if (level_search_impl<MT, true, false, 1>(a.gv, lds, qa2, closest, slot, level, a.top_limit, L, unused_queue, score, wc) !=
		    LEVEL_OK) {
			if (lane == 0) {
				a.node_status[node] = 1;
				atomicExch(&a.counters[3], 1u);
			}
			return;
		}
		if constexpr (E == 0) { // refine_ reads the list where it lies (every entry carries the "expanded" mark by now)
			for (int i = lane; i < L.size; i += 64)
				L.s[i] &= ~EXPANDED_BIT;
			lds.cand_d = L.d;
			lds.cand_s = L.s;
		} else {
			L.dump(lds.cand_d, lds.cand_s);
		}
		wave_sync();
		const int kept = refine_candidates<MT, NCH, R>(a.gv, lds, L.size, a.gv.M, wc); // needed = M on every level (:3665)
		// connect_new_node_: the node's own (blank) list
		uint32_t *mine = src == EMPTY_SLOT ? a.gv.list_ptr(slot, level)
		                                   : a.parked + (size_t)node * a.parked_stride + (size_t)level * a.list_cap_max;
		const uint32_t cap = a.gv.list_cap(level);
		for (uint32_t i = lane; i < cap; i += 64)
			mine[i] = i < (uint32_t)kept ? lds.kept_s[i] : EMPTY_SLOT;
		// reverse-link requests, in selection order
		for (int i = lane; i < kept; i += 64) {
			const uint32_t t = lds.kept_s[i];
			req_l[n_req + i] =
			    t == slot ? EMPTY_SLOT : (level == 0 ? t : a.gv.list_id_base + a.gv.upper_off[t] + (level - 1));
			req_dd[n_req + i] = lds.kept_d[i];
		}
		n_req += kept;
		closest = lds.kept_s[0];
		wave_sync();
	}
	uint32_t base = 0;
	if (lane == 0) {
		base = atomicAdd(&a.counters[0], n_req);
		a.node_status[node] = 0;
		atomicAdd(&a.work_stats[0], (unsigned long long)wc.distances);
		atomicAdd(&a.work_stats[1], (unsigned long long)wc.cycles);
	}
	base = read_lane(base, 0);
	for (uint32_t i = lane; i < n_req; i += 64) {
		const uint32_t idx = base + i;
		if (idx < a.req_capacity) {
			a.req_list[idx] = req_l[i];
			a.req_src[idx] = slot | (src == EMPTY_SLOT ? 0u : REUSED_SOURCE);
			a.req_d[idx] = req_dd[i];
		}
	}
}

// =========================================================================================================
// Bulk build, phase B — group the batch's reverse-link requests by target list, then one wave per touched list
// applies them in ascending source order (the reference's reconnect step per incoming link).
// =========================================================================================================
struct LinkArgs {
	GraphView gv;
	uint32_t *req_list;
	uint32_t *req_src;
	float *req_d;
	uint32_t *req_rank;    // rank of a request within its list (arrival order, arbitrary)
	uint32_t *counters;    // see BuildArgs
	uint32_t *list_count;  // per list id: requests this batch (zero between batches)
	uint32_t *list_offset; // per list id: start in sorted_*
	uint32_t *touched;     // list ids with >= 1 request
	uint32_t *sorted_src;
	float *sorted_d;
	const uint32_t *list_owner; // upper list -> owning slot
	const uint32_t *upper_off;
	uint32_t hash_log2;
	uint32_t list_cap_max;
	unsigned long long *work_stats; // [2] += distances computed by the link repairs
};

#ifdef VSS_ENGINE_TU // plain kernels are defined once, in the engine's translation unit
// Reused slots, before phase A: blank every list of the node (update() zeroes the node tape, index.hpp:2837-2840).
// After phase A: move the parked new lists in.  One wave per batch node; appended nodes are skipped.
__global__ __launch_bounds__(64) void k_reuse_lists(BuildArgs a, int commit) {
	const uint32_t node = blockIdx.x;
	if (a.row_src[node] == EMPTY_SLOT)
		return;
	const uint32_t slot = a.row_slot[node];
	const int target = a.levels[slot];
	const int top = commit ? (target < a.max_level ? target : a.max_level) : target;
	for (int level = 0; level <= top; ++level) {
		uint32_t *lp = a.gv.list_ptr(slot, level);
		const uint32_t *from = a.parked + (size_t)node * a.parked_stride + (size_t)level * a.list_cap_max;
		for (uint32_t i = threadIdx.x; i < a.gv.list_cap(level); i += 64)
			lp[i] = commit ? from[i] : EMPTY_SLOT;
	}
}

// index.remove(): key := free key for a batch of slots (index_dense.hpp:1228-1255)
__global__ void k_mark_removed(int64_t *keys, const uint32_t *slots, uint32_t n) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
		keys[slots[i]] = FREE_KEY;
}

// ---------------------------------------------------------------------------------------------------------
// vss_compact on the device: drop the tombstoned nodes, renumber the survivors densely in slot order, remove the
// links that pointed at dropped nodes (the others keep their order).  The host computes the two slot maps (it owns the
// key mirror) and the new offsets of the upper lists; the kernels move the data.
struct CompactArgs {
	const uint32_t *src_of; // new slot -> old slot          [live]
	const uint32_t *remap;  // old slot -> new slot or EMPTY [count]
	uint32_t live;
	uint32_t V, M, M0;
	const float4 *vectors;
	float4 *staging; // rows of one chunk on their way down
	const uint32_t *links0, *links_up, *upper_off;
	const uint8_t *levels;
	const int64_t *keys;
	uint32_t *links0_new, *links_up_new, *list_owner_new;
	const uint32_t *upper_off_new; // per new slot (host-computed prefix sums)
	uint8_t *levels_new;
	int64_t *keys_new;
};

// rows [first, first + n) of the NEW numbering, gathered from their old places into `staging` (the rows move towards
// lower slots, so a chunk is staged and then copied over its destination: later chunks read only rows behind it)
__global__ __launch_bounds__(256) void k_compact_rows(CompactArgs a, uint32_t first, uint32_t n) {
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
	for (uint32_t i = wave; i < n; i += n_waves) {
		const float4 *src = a.vectors + (size_t)a.src_of[first + i] * a.V;
		float4 *dst = a.staging + (size_t)i * a.V;
		for (uint32_t c = lane; c < a.V; c += 64)
			dst[c] = src[c];
	}
}

// one wave per surviving node: key, level, and every list filtered through `remap`
__global__ __launch_bounds__(256) void k_compact_links(CompactArgs a) {
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
	for (uint32_t t = wave; t < a.live; t += n_waves) {
		const uint32_t s = a.src_of[t];
		const int level = a.levels[s];
		if (lane == 0) {
			a.keys_new[t] = a.keys[s];
			a.levels_new[t] = (uint8_t)level;
		}
		for (int l = 0; l <= level; ++l) {
			const uint32_t cap = l ? a.M : a.M0;
			const uint32_t *from = l ? a.links_up + ((size_t)a.upper_off[s] + (l - 1)) * a.M : a.links0 + (size_t)s * a.M0;
			uint32_t *to = l ? a.links_up_new + ((size_t)a.upper_off_new[t] + (l - 1)) * a.M : a.links0_new + (size_t)t * a.M0;
			if (l && lane == 0)
				a.list_owner_new[a.upper_off_new[t] + (l - 1)] = t;
			uint32_t kept = 0;
			for (uint32_t off = 0; off < cap; off += 64) {
				uint32_t id = off + lane < cap ? from[off + lane] : EMPTY_SLOT;
				if (id != EMPTY_SLOT)
					id = a.remap[id];
				const unsigned long long m = __ballot(id != EMPTY_SLOT);
				if (id != EMPTY_SLOT)
					to[kept + __popcll(m & lanes_below((int)lane))] = id;
				kept += __popcll(m);
			}
			for (uint32_t i = kept + lane; i < cap; i += 64)
				to[i] = EMPTY_SLOT;
		}
	}
}

__global__ void k_link_count(LinkArgs a) {
	const uint32_t n = a.counters[0];
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t l = a.req_list[i];
		if (l == EMPTY_SLOT)
			continue;
		const uint32_t c = atomicAdd(&a.list_count[l], 1u);
		a.req_rank[i] = c;
		if (c == 0)
			a.touched[atomicAdd(&a.counters[1], 1u)] = l;
	}
}

__global__ void k_link_alloc(LinkArgs a) {
	const uint32_t n = a.counters[1];
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t l = a.touched[i];
		a.list_offset[l] = atomicAdd(&a.counters[2], a.list_count[l]);
	}
}

__global__ void k_link_scatter(LinkArgs a) {
	const uint32_t n = a.counters[0];
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t l = a.req_list[i];
		if (l == EMPTY_SLOT)
			continue;
		const uint32_t p = a.list_offset[l] + a.req_rank[i];
		a.sorted_src[p] = a.req_src[i];
		a.sorted_d[p] = a.req_d[i];
	}
}

#endif // VSS_ENGINE_TU

// ---------------------------------------------------------------------------------------------------------
// vss_compact, step 1 — "for every bottom level node, determine its parent cluster" (index_gt::compact, index.hpp:3431-3446):
// cluster(s) = search_for_one_(vector of s, entry, max_level, 0) = the node the greedy descent through levels
// max_level .. 1 lands on.  One wave per live node (grid-stride); tombstoned nodes are dropped by the compaction and get
// no cluster.  Read-only on the graph; the distances are the production wave-order ones, so the oracle's mirror
// (compact_reordering) finds the same clusters bit for bit.
struct ClusterArgs {
	GraphView gv;
	uint32_t count;      // slots 0 .. count-1
	uint32_t entry;
	int max_level;
	uint32_t list_cap_max;
	uint32_t *cluster;   // out: per slot (EMPTY_SLOT for tombstones)
	unsigned long long *work_stats; // [0] += distances computed, [1] += nodes expanded
};

template <int MT, int NCH, int R>
__global__ __launch_bounds__(64) void k_node_clusters(ClusterArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = lane_id();
	WaveLds lds;
	carve_lds(lds, smem, 4, a.gv.sp.V, a.list_cap_max, 16); // no visited set needed by the descent (a token 16 cells)
	const SoloScorer<MT, NCH, R> score;
	WorkCounters wc = {};
	for (uint32_t slot = blockIdx.x; slot < a.count; slot += gridDim.x) {
		if (a.gv.keys[slot] == FREE_KEY) {
			if (lane == 0)
				a.cluster[slot] = EMPTY_SLOT;
			continue;
		}
		stage_row(lds.q, a.gv.sp.vectors + (size_t)slot * a.gv.sp.V, a.gv.sp.V);
		const float qa2 = MT == 1 ? wave_query_norm(a.gv.sp, lds.q) : 0.f;
		const uint32_t c = descend<MT>(a.gv, lds, qa2, a.entry, a.max_level, 0, score, wc);
		if (lane == 0)
			a.cluster[slot] = c;
		wave_sync();
	}
	if (lane == 0 && a.work_stats) {
		atomicAdd(&a.work_stats[0], (unsigned long long)wc.distances);
		atomicAdd(&a.work_stats[1], (unsigned long long)wc.cycles);
	}
}

#ifndef VSS_PHASE_B_WAVES_PER_EU
#define VSS_PHASE_B_WAVES_PER_EU 1 // lower bound handed to the register allocator (A/B builds)
#endif
template <int MT, int NCH, int R>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(VSS_PHASE_B_WAVES_PER_EU, 8))) void k_build_phase_b(LinkArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = lane_id();
	const uint32_t n_touched = a.counters[1];
	WaveLds lds;
	const uint32_t cand_cap = a.list_cap_max + 1;
	carve_lds(lds, smem, a.hash_log2, a.gv.sp.V, a.list_cap_max, cand_cap);
	WorkCounters wc = {};
	for (uint32_t t = blockIdx.x; t < n_touched; t += gridDim.x) {
		const uint32_t lid = a.touched[t];
		const uint32_t n_in = a.list_count[lid];
		const uint32_t in_off = a.list_offset[lid];
		uint32_t slot;
		int level;
		if (lid < a.gv.list_id_base) {
			slot = lid;
			level = 0;
		} else {
			const uint32_t u = lid - a.gv.list_id_base;
			slot = a.list_owner[u];
			level = 1 + (int)(u - a.upper_off[slot]);
		}
		const uint32_t cap = a.gv.list_cap(level);
		uint32_t *lp = a.gv.list_ptr(slot, level);
		// current list -> kept_s (the working copy), its distances to `slot` are computed lazily
		int cur = 0;
		for (uint32_t off = 0; off < cap; off += 64) {
			uint32_t id = (off + lane < cap) ? lp[off + lane] : EMPTY_SLOT;
			unsigned long long m = __ballot(id != EMPTY_SLOT);
			if (id != EMPTY_SLOT)
				lds.kept_s[cur + __popcll(m & lanes_below(lane))] = id;
			cur += __popcll(m);
		}
		wave_sync();
		bool have_d = false;
		bool staged = false;
		float n2 = 0.f;
		uint32_t last_src = 0;
		for (uint32_t k = 0; k < n_in; ++k) {
			// next incoming link in ascending source order: the smallest source > last_src (sources are unique)
			uint32_t best = EMPTY_SLOT;
			float best_d = 0.f;
			bool best_reused = false;
			for (uint32_t off = 0; off < n_in; off += 64) {
				uint32_t s = EMPTY_SLOT;
				float d = 0.f;
				bool reused = false;
				if (off + lane < n_in) {
					const uint32_t raw = a.sorted_src[in_off + off + lane];
					s = raw & ~REUSED_SOURCE;
					reused = (raw & REUSED_SOURCE) != 0;
					d = a.sorted_d[in_off + off + lane];
					if (k > 0 && s <= last_src)
						s = EMPTY_SLOT;
				}
				uint32_t m = s;
				for (int o = 32; o >= 1; o >>= 1) {
					uint32_t other = __shfl_xor(m, o);
					m = other < m ? other : m;
				}
				if (m < best) {
					const int who = __builtin_ctzll(__ballot(s == m));
					best = m;
					best_d = __shfl(d, who);
					best_reused = __shfl((int)reused, who) != 0;
				}
			}
			last_src = best;
			if ((uint32_t)cur < cap) { // room left: plain append (index.hpp:3701-3704)
				if (lane == 0) {
					lds.kept_s[cur] = best;
					lds.kept_d[cur] = best_d;
				}
				cur++;
				if (best_reused)
					have_d = false;
				wave_sync();
				continue;
			}
			// the list is full: rebuild it from {new} + successors with refine_ (index.hpp:3706-3719)
			if (!staged) {
				stage_row(lds.q, a.gv.sp.vectors + (size_t)slot * a.gv.sp.V, a.gv.sp.V);
				n2 = MT == 1 ? wave_query_norm(a.gv.sp, lds.q) : 0.f;
				staged = true;
			}
			if (!have_d) {
				wave_distances<MT, NCH, R>(a.gv.sp, lds.q, n2, lds.kept_s, cur, lds.kept_d);
				wc.distances += cur;
				have_d = true;
			}
			// sorted_buffer_gt::insert_reserved order: the new link first, then the successors in list order,
			// each placed BEFORE entries of equal distance -> rank = #smaller + #equal inserted later
			const int total = cur + 1;
			for (int i = lane; i < total; i += 64) {
				const float di = i == 0 ? best_d : lds.kept_d[i - 1];
				const uint32_t si = i == 0 ? best : lds.kept_s[i - 1];
				int rank = 0;
				for (int j = 0; j < total; ++j) {
					const float dj = j == 0 ? best_d : lds.kept_d[j - 1];
					rank += (dj < di) || (dj == di && j > i);
				}
				lds.cand_d[rank] = di;
				lds.cand_s[rank] = si;
			}
			wave_sync();
			cur = refine_candidates<MT, NCH, R>(a.gv, lds, total, (int)cap, wc);
			if (best_reused) // its distance above came from the NEW vector; later rebuilds measure the stored (old) one
				have_d = false;
			// refine_ left the selection (with its distances to `slot`) in kept_s / kept_d; lds.q2 was clobbered only
		}
		for (uint32_t i = lane; i < cap; i += 64)
			lp[i] = i < (uint32_t)cur ? lds.kept_s[i] : EMPTY_SLOT;
		if (lane == 0)
			a.list_count[lid] = 0; // leave the counter array clean for the next batch
		wave_sync();
	}
	if (lane == 0 && wc.distances)
		atomicAdd(&a.work_stats[2], (unsigned long long)wc.distances);
}

} // namespace vss
