// wave_primitives.h — per-wavefront building blocks of the HNSW kernels (gfx950, wave64).
//
// One wavefront (64 lanes) owns one query / one node being inserted / one neighbour list being repaired.
// Everything here is wave-synchronous: the LDS regions are private to the walking wave and `wave_sync()` (wait for the
// wave's own outstanding memory operations + a compiler barrier; a wave's LDS traffic is performed in issue order)
// orders its cross-lane LDS traffic.  Search teams (hnsw_kernels.h) add helper waves that only score rows; those
// meet the walking wave at real workgroup barriers.
//
//   * WaveList<E>   a sorted candidate list living in registers: lane l holds E consecutive slots, the list right-aligned (round 6)
//                   — replaces usearch's `top` sorted_buffer_gt + `next` max_heap_gt (index.hpp:783-917, 620-773)
//   * VisitedSet    an exact open-addressing set in LDS — replaces growing_hash_set_gt (index.hpp:1018-1144)
//   * wave_distances  distances from one staged query to a handful of rows, rows read as coalesced float4
//                   streams, reduced with an xor-butterfly — replaces metric_punned_t (index_plugins.hpp:977-1053)
//
// Summation order ("wave order", restated on the CPU by oracle/hnsw_oracle.cpp dist_wave_order): a row of
// V = ceil(dim/4) float4 chunks is handled by G = min(64, pow2ceil(V)) lanes; lane g accumulates chunks
// g, g+G, g+2G, ... component by component with fmaf, then the G partial sums are combined by
// acc += shfl_xor(acc, off) for off = G/2 ... 1.  Compiled with -ffp-contract=off so nothing else fuses.
//
// The metric (MT: 0 l2sq, 1 cosine, 2 ip), the chunks per lane (NCH) and the list registers (E) are template
// parameters: a run-time metric switch inside the accumulate loop made hipcc emit branchy code that copied the whole
// accumulator file at every merge point (measured 5.4k cycles for 200 FMAs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "visited_compact.h"

namespace vss {

constexpr uint32_t EMPTY_SLOT = 0xFFFFFFFFu;
constexpr uint32_t EXPANDED_BIT = 0x80000000u;
constexpr int MAX_LIST_REGS = 8; // 64 * 8 = 512 entries: the largest ef / ef_construction a register list holds

// Correctly rounded f32 square root (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt applies to sqrtf; the
// __fsqrt_rn intrinsic of ROCm 7.2 maps to the NATIVE, ~1 ulp, square root and made cosine distances of vectors with
// non-unit norms differ from the reference by an ulp or two — found by the degenerate-data fuzz).
__device__ __forceinline__ float vss_sqrt(float x) {
	return __builtin_sqrtf(x);
}

__device__ __forceinline__ int lane_id() {
	return threadIdx.x & 63;
}
__device__ __forceinline__ void wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); // s_waitcnt vmcnt(0) lgkmcnt(0); no s_barrier
	__builtin_amdgcn_wave_barrier();
}
// The same for traffic that is known to be LDS only: waits for the wave's LDS operations (s_waitcnt lgkmcnt(0)) and leaves
// its global loads in flight — list requests and touches issued ahead of time stay asynchronous across it.  (Flat
// instructions on LDS addresses count on lgkmcnt as well.)  Not for lists / visited sets that live in HBM.
__device__ __forceinline__ void lds_sync() {
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
	__builtin_amdgcn_wave_barrier();
}
// Workgroup barrier for hand-overs through LDS: what this wave wrote to LDS is visible to the waves it meets; its global
// loads are NOT waited for (__syncthreads() drains them: s_waitcnt vmcnt(0) ahead of the s_barrier).
__device__ __forceinline__ void lds_barrier() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
	__builtin_amdgcn_s_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ unsigned long long lanes_below(int lane) {
	return (1ull << lane) - 1ull;
}
__device__ __forceinline__ int uniform(int v) {
	return __builtin_amdgcn_readfirstlane(v);
}
// value of `v` in lane `src` (src must be wave-uniform): v_readlane_b32, no LDS crossbar round trip
__device__ __forceinline__ float read_lane(float v, int src) {
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(src)));
}
__device__ __forceinline__ uint32_t read_lane(uint32_t v, int src) {
	return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src));
}
// lane i <- lane i-1, lane 0 <- `first`: one DPP move (wave_shr:1, gfx9 family incl. gfx950)
__device__ __forceinline__ float shift_up_one(float first, float v) {
	return __int_as_float(
	    __builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t shift_up_one(uint32_t first, uint32_t v) {
	return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xf, 0xf, false);
}

// lane i <- lane i+1, lane 63 <- `last`: one DPP move (wave_shl:1)
__device__ __forceinline__ float shift_down_one(float last, float v) {
	return __int_as_float(
	    __builtin_amdgcn_update_dpp(__float_as_int(last), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t shift_down_one(uint32_t last, uint32_t v) {
	return (uint32_t)__builtin_amdgcn_update_dpp((int)last, (int)v, 0x130, 0xf, 0xf, false);
}

// v from lane (lane ^ OFF), all 64 lanes active.  Entirely on the VALU (DPP / permlane swaps), no LDS-crossbar round trip:
//   1, 2: quad_perm   4: row_half_mirror o quad_perm[3,2,1,0]   8: row_ror:8   16: v_permlane16_swap   32: v_permlane32_swap
template <int OFF>
__device__ __forceinline__ float lane_xor(float v) {
	const int iv = __float_as_int(v);
	if constexpr (OFF == 1) {
		return __int_as_float(__builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xf, 0xf, true));
	} else if constexpr (OFF == 2) {
		return __int_as_float(__builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xf, 0xf, true));
	} else if constexpr (OFF == 4) {
		const int t = __builtin_amdgcn_update_dpp(0, iv, 0x141, 0xf, 0xf, true); // lane i <- i ^ 7 (within 8)
		return __int_as_float(__builtin_amdgcn_update_dpp(0, t, 0x1B, 0xf, 0xf, true)); // then i ^ 3  => i ^ 4
	} else if constexpr (OFF == 8) {
		return __int_as_float(__builtin_amdgcn_update_dpp(0, iv, 0x128, 0xf, 0xf, true)); // row_ror:8 within 16 lanes
	} else if constexpr (OFF == 16) {
		// odd rows of the first operand swap with even rows of the second: {r0,r0,r2,r2} and {r1,r1,r3,r3}
		const auto r = __builtin_amdgcn_permlane16_swap((unsigned)iv, (unsigned)iv, false, false);
		return __int_as_float((int)((lane_id() & 16) ? r[0] : r[1]));
	} else {
		static_assert(OFF == 32, "lane_xor offset");
		// upper half of the first operand swaps with the lower half of the second: {lo,lo} and {hi,hi}
		const auto r = __builtin_amdgcn_permlane32_swap((unsigned)iv, (unsigned)iv, false, false);
		return __int_as_float((int)((lane_id() & 32) ? r[0] : r[1]));
	}
}
// run-time offset (power of two <= 32), wave-uniform
__device__ __forceinline__ float lane_xor_dyn(float v, uint32_t off) {
	switch (off) {
	case 32:
		return lane_xor<32>(v);
	case 16:
		return lane_xor<16>(v);
	case 8:
		return lane_xor<8>(v);
	case 4:
		return lane_xor<4>(v);
	case 2:
		return lane_xor<2>(v);
	default:
		return lane_xor<1>(v);
	}
}

// ------------------------------------------------------------------------------------------------------
// WaveList
// ------------------------------------------------------------------------------------------------------
// Round 6: BLOCKED, RIGHT-ALIGNED layout.  Rounds 1-5 kept entry p at (lane p % 64, register p / 64): a sorted insert then cost
// one ballot + popcount per register for the position and, per register, two DPP moves, two v_readlane carries and four selects
// — ~140 dependent-ish instructions for the 8-register list, with half a dozen VALU -> SALU -> branch hops — and it was the
// walker's instruction stream, not memory, that bounded every regime short of the streaming one.  Now:
//   * lane l holds the E CONSECUTIVE physical slots l*E .. l*E+E-1 (register r = slot l*E + r), and the list is RIGHT-aligned:
//     list position p lives in physical slot off + p, off = 64 E - limit.  Slots below `off` hold -inf (and the expanded mark),
//     slots from off + size on hold +inf (and the expanded mark): every comparison works on all 64 E slots, no validity masks.
//   * insert(nd): m_r = (d[r] < nd) per register (E compares into E lane masks); an entry moves one slot up iff it is NOT less
//     than nd, so new[r] = m_r ? old[r] : (m_{r-1} ? NEW : old[r-1]), where register -1 of a lane is register E-1 of the lane
//     below it: ONE DPP move per array for the whole list, and its mask is (m_{E-1} << 1) | 1 — scalar arithmetic.  The entry
//     that leaves the last slot falls off the end: exactly sorted_buffer_gt::insert's eviction (index.hpp:880-891), for free.
//     No position is ever computed: 5 vector instructions per register, two DPP moves, no ballot, no branch in the common case.
//   * "new before equal" (lower_bound) is the strict `<` of the compare; a candidate no entry is >= to is rejected by the same
//     masks (nothing moves).
//   * the first unexpanded entry: per lane a select chain over its E registers, one ballot, three v_readlane.
// Same semantics as before, entry for entry (the oracle's kernel mode restates the LIST, not its layout): ids, distance bits,
// graph bytes and counters are unchanged — every collected parity test holds it to that.  The batched merge of rounds 2-5
// (ballots per candidate and register + an LDS staging round trip) is gone: sequential inserts are cheaper now at any count.
// Distances that are not finite (NaN, +-inf: only from such inputs) take insert_slow(), which restates the old positional insert.
template <int E>
struct WaveList {
	static_assert(E == 1 || E == 2 || E == 4 || E == 8 || E == 16, "list registers (16: the RegQueue next to an 8-register list)");
	static constexpr bool can_merge = false;
	static constexpr int regs = E;
	static constexpr int prefetch_slots = E <= 4 ? 2 : 1; // neighbour lists kept in flight (ListCache)
	static constexpr int LOG_E = E == 1 ? 0 : E == 2 ? 1 : E == 4 ? 2 : E == 8 ? 3 : 4;
	float d[E];
	uint32_t s[E]; // bit 31 = "already expanded" (set on every padding slot as well)
	int size;      // wave-uniform
	int limit;     // wave-uniform capacity (<= 64 * E)
	int off;       // wave-uniform: physical slot of list position 0 (= 64 E - limit)

	__device__ __forceinline__ void reset(int lim) {
		limit = uniform(lim); // (readfirstlane: tells the compiler these live in scalar registers)
		size = 0;
		off = uniform(64 * E - lim);
		const int base = lane_id() * E;
#pragma unroll
		for (int r = 0; r < E; ++r) {
			d[r] = base + r < off ? -__builtin_inff() : __builtin_inff();
			s[r] = EXPANDED_BIT;
		}
	}

	// One element enters, every entry that is not smaller moves one slot up, the entry in the last slot falls off.
	// `ok` (wave-uniform): false = nothing happens (the caller's radius test folded into the lane masks: the accept loop below
	// stays ONE basic block — no branch per candidate, and the compiler updates the registers in place).
	// FRONT (NaN or -inf, which the padding cannot order; only from such inputs, never on finite data): the element goes to
	// list position 0 — where the positional insert of rounds 1-5 put it (no entry compares smaller) — by slot number
	// instead of by distance.
	template <bool FRONT>
	__device__ __forceinline__ void place(float nd, uint32_t ns, bool ok = true) {
		const int first = off - lane_id() * E; // registers below this index (if any) lie in front of the list
		// what enters register 0 of a lane: register E-1 of the lane below (lane 0: the -inf padding in front of everything)
		const float up_d = shift_up_one(-__builtin_inff(), d[E - 1]);
		const uint32_t up_s = shift_up_one(EXPANDED_BIT, s[E - 1]);
		bool stays = !ok || (FRONT ? E - 1 < first : d[E - 1] < nd); // is this slot's entry smaller than the new element?
		// every entry is smaller (the last slot holds the largest): rejected — it would land at `limit`; nothing moves
		const uint32_t last_lanes = (uint32_t)(__ballot(stays) >> 32);
		const bool rejected = (last_lanes >> 31) != 0u;
#pragma unroll
		for (int r = E - 1; r >= 0; --r) {
			// the slot BELOW this one: smaller as well?  (if it is and this one is not, the new element enters here)
			const bool below_stays =
			    !ok || (r > 0 ? (FRONT ? r - 1 < first : d[r > 0 ? r - 1 : 0] < nd) : (FRONT ? -1 < first : up_d < nd));
			const float prev_d = r > 0 ? d[r > 0 ? r - 1 : 0] : up_d;
			const uint32_t prev_s = r > 0 ? s[r > 0 ? r - 1 : 0] : up_s;
			d[r] = stays ? d[r] : (below_stays ? nd : prev_d);
			s[r] = stays ? s[r] : (below_stays ? ns : prev_s);
			stays = below_stays;
		}
		size = uniform((size < limit && !rejected) ? size + 1 : size);
	}

	// sorted_buffer_gt::insert(element, limit), index.hpp:880-891: position = lower_bound (the new element goes BEFORE equal
	// distances); rejected if it would land at `limit` (then nothing changes); the last entry falls off when full.
	template <bool SKIP = false> // (the flavour switch of round 5's lane-major list; nothing to choose here)
	__device__ __forceinline__ void insert(float nd, uint32_t ns) {
		if (__builtin_expect(!(nd > -__builtin_inff()), 0)) // wave-uniform; NaN / -inf
			place<true>(nd, ns);
		else
			place<false>(nd, ns);
	}

	// The accept phase of an expansion (search_to_find_in_base_ / search_to_insert_, index.hpp:3981-3992 / 3905-3913, without
	// tombstones): the lanes named in `pass` hold one fresh (distance, slot) pair each; in lane order, each is inserted iff the
	// list is not full or it beats the radius AT THAT MOMENT.  `radius` comes back as the last slot's distance — the radius of
	// a full list, +inf while it is filling (callers test `size < limit ||` first).
	__device__ __forceinline__ void accept(float cd, uint32_t cs, unsigned long long pass, float &radius) {
		const bool mine = (pass >> lane_id()) & 1ull;
		if (__builtin_expect(__ballot(mine && !(cd > -__builtin_inff())) != 0ull, 0)) { // a NaN / -inf among them: the careful way
			while (pass) {
				const int j = __builtin_ctzll(pass);
				pass &= pass - 1;
				const float dj = read_lane(cd, j);
				if (size < limit || dj < radius) {
					insert(dj, read_lane(cs, j));
					radius = last_distance();
				}
			}
			return;
		}
		// Every distance is finite from here on.  Whether a candidate enters is told by the list itself — the LAST slot holds the
		// radius of a full list and +inf padding while the list is filling, so "list not full, or d < radius" is
		// `!(last <= d)` — which takes the radius (a v_readlane after the selects of the previous candidate) and the size out
		// of the loop-carried chain: an iteration depends on the previous one through the list registers only.
		const int entering = __popcll(pass);
		while (pass) {
			const int j = __builtin_ctzll(pass);
			pass &= pass - 1;
			place_finite(read_lane(cd, j), read_lane(cs, j));
		}
		size = uniform(size + entering < limit ? size + entering : limit); // (filling: each one entered; full: it stays full)
		radius = last_distance();
	}

	// place<false> for a finite distance, the radius test included: nothing moves when the last slot's entry is <= nd.
	__device__ __forceinline__ void place_finite(float nd, uint32_t ns) {
		// (lane 0 gets zeros from the DPP move — nothing has to be materialised for it: its "slot below" is the padding in front
		//  of everything, which the lane test stands for, and the values are never selected there)
		const float up_d = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d[E - 1]), 0x138, 0xf, 0xf, true));
		const uint32_t up_s = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s[E - 1], 0x138, 0xf, 0xf, true);
		const bool rejected = ((uint32_t)(__ballot(d[E - 1] <= nd) >> 32) >> 31) != 0u; // wave-uniform: lane 63's compare
		bool stays = rejected || d[E - 1] < nd;
#pragma unroll
		for (int r = E - 1; r >= 0; --r) {
			const bool below_stays = rejected || (r > 0 ? d[r > 0 ? r - 1 : 0] < nd : (up_d < nd || lane_id() == 0));
			const float prev_d = r > 0 ? d[r > 0 ? r - 1 : 0] : up_d;
			const uint32_t prev_s = r > 0 ? s[r > 0 ? r - 1 : 0] : up_s;
			d[r] = stays ? d[r] : (below_stays ? nd : prev_d);
			s[r] = stays ? s[r] : (below_stays ? ns : prev_s);
			stays = below_stays;
		}
	}

	__device__ __forceinline__ void get(int pos, float &od, uint32_t &os) const {
		const int ph = uniform(off + pos);
		const int reg = ph & (E - 1);
		float vd = d[0];
		uint32_t vs = s[0];
#pragma unroll
		for (int r = 1; r < E; ++r) { // (reg is wave-uniform: scalar conditions)
			vd = reg == r ? d[r] : vd;
			vs = reg == r ? s[r] : vs;
		}
		od = read_lane(vd, ph >> LOG_E);
		os = read_lane(vs, ph >> LOG_E);
	}

	__device__ __forceinline__ float last_distance() const {
		if (size == limit) // (the radius of a full list: the last physical slot)
			return read_lane(d[E - 1], 63);
		float od;
		uint32_t os;
		get(size - 1, od, os);
		return od;
	}

	// The first unexpanded entry (-1: none): its position, distance and slot word.  Per lane a select chain over its E
	// registers (padding slots carry the expanded mark), one ballot for the lane, three v_readlane.
	__device__ __forceinline__ int first_unexpanded_entry(float &od, uint32_t &os) const {
		int idx = E;
		float fd = 0.f;
		uint32_t fs = 0;
#pragma unroll
		for (int r = E - 1; r >= 0; --r) {
			const bool u = (int)s[r] >= 0; // bit 31 clear
			idx = u ? r : idx;
			fd = u ? d[r] : fd;
			fs = u ? s[r] : fs;
		}
		const unsigned long long m = __ballot(idx < E);
		if (!m)
			return -1;
		const int l = __builtin_ctzll(m);
		od = read_lane(fd, l);
		os = read_lane(fs, l);
		return l * E + (int)read_lane((uint32_t)idx, l) - off;
	}
	__device__ __forceinline__ int first_unexpanded() const {
		float od;
		uint32_t os;
		return first_unexpanded_entry(od, os);
	}
	// The best TWO unexpanded entries' slot words in one pass (the look-ahead's list requests); returns how many there are.
	__device__ __forceinline__ int first_two_unexpanded(uint32_t &s1, uint32_t &s2) const {
		int n1 = 0; // unexpanded entries of this lane, saturating at 2
		uint32_t a = 0, b = 0; // their slot words (first, second)
#pragma unroll
		for (int r = E - 1; r >= 0; --r) {
			const bool u = (int)s[r] >= 0;
			b = u ? a : b;
			a = u ? s[r] : a;
			n1 = u ? (n1 < 2 ? n1 + 1 : 2) : n1;
		}
		const unsigned long long m = __ballot(n1 > 0);
		if (!m)
			return 0;
		const int l1 = __builtin_ctzll(m);
		s1 = read_lane(a, l1);
		if ((int)read_lane((uint32_t)n1, l1) > 1) {
			s2 = read_lane(b, l1);
			return 2;
		}
		const unsigned long long rest = m & (m - 1);
		if (!rest)
			return 1;
		s2 = read_lane(a, __builtin_ctzll(rest));
		return 2;
	}

	// first_two_unexpanded() that also tells the FIRST one's distance — the pipelined level search finds the best unexpanded
	// entry once, while the walker has nothing else to do, for the look-ahead's list requests AND for the next pick
	__device__ __forceinline__ int first_two_unexpanded_entry(float &d1, uint32_t &s1, uint32_t &s2) const {
		int n1 = 0;
		uint32_t a = 0, b = 0;
		float fd = 0.f;
#pragma unroll
		for (int r = E - 1; r >= 0; --r) {
			const bool u = (int)s[r] >= 0;
			b = u ? a : b;
			a = u ? s[r] : a;
			fd = u ? d[r] : fd;
			n1 = u ? (n1 < 2 ? n1 + 1 : 2) : n1;
		}
		const unsigned long long m = __ballot(n1 > 0);
		if (!m)
			return 0;
		const int l1 = __builtin_ctzll(m);
		s1 = read_lane(a, l1);
		d1 = read_lane(fd, l1);
		if ((int)read_lane((uint32_t)n1, l1) > 1) {
			s2 = read_lane(b, l1);
			return 2;
		}
		const unsigned long long rest = m & (m - 1);
		if (!rest)
			return 1;
		s2 = read_lane(a, __builtin_ctzll(rest));
		return 2;
	}

	// the first unexpanded entry behind position `pos` (-1: none)
	__device__ __forceinline__ int next_unexpanded(int pos) const {
		const int base = lane_id() * E, ph0 = off + pos;
		int idx = E;
#pragma unroll
		for (int r = E - 1; r >= 0; --r)
			idx = ((int)s[r] >= 0 && base + r > ph0) ? r : idx;
		const unsigned long long m = __ballot(idx < E);
		if (!m)
			return -1;
		const int l = __builtin_ctzll(m);
		return l * E + (int)read_lane((uint32_t)idx, l) - off;
	}

	__device__ __forceinline__ void mark_expanded(int pos) {
		const int t = off + pos - lane_id() * E; // which register of this lane, if any
#pragma unroll
		for (int r = 0; r < E; ++r)
			s[r] |= t == r ? EXPANDED_BIT : 0u;
	}

	// drop entry 0: every entry moves one position down (the RegQueue of searches over tombstones / a predicate)
	__device__ __forceinline__ void remove_first() {
		const float down_d = shift_down_one(__builtin_inff(), d[0]); // register 0 of the lane above enters register E-1
		const uint32_t down_s = shift_down_one(EXPANDED_BIT, s[0]);
		const int base = lane_id() * E;
#pragma unroll
		for (int r = 0; r < E; ++r) {
			const float nd = r + 1 < E ? d[r + 1 < E ? r + 1 : 0] : down_d;
			const uint32_t ns = r + 1 < E ? s[r + 1 < E ? r + 1 : 0] : down_s;
			const bool in_list = base + r >= off; // (the padding below the list stays what it is)
			d[r] = in_list ? nd : d[r];
			s[r] = in_list ? ns : s[r];
		}
		size = uniform(size - 1);
	}

	// dump the list (ascending) into LDS arrays
	__device__ __forceinline__ void dump(float *out_d, uint32_t *out_s) const {
		const int base = lane_id() * E - off;
#pragma unroll
		for (int r = 0; r < E; ++r) {
			const int pos = base + r;
			if (pos >= 0 && pos < size) {
				out_d[pos] = d[r];
				out_s[pos] = s[r] & ~EXPANDED_BIT;
			}
		}
	}
};

// ------------------------------------------------------------------------------------------------------
// MemList / CandQueue — the same lists in memory (LDS or HBM, addressed through generic pointers) for limits the
// register list cannot hold (ef_search, k or ef_construction above 64 * MAX_LIST_REGS) and for the candidate queue of
// searches over tombstones / a predicate, which the reference keeps unbounded (`next` heap, index.hpp:3981-3992).
// Same semantics as WaveList, entry for entry (sorted_buffer_gt::insert: new element BEFORE equal distances); costs
// O(size / 64) per operation instead of O(E), which only the rare large-limit searches pay.
// ------------------------------------------------------------------------------------------------------
struct MemList {
	static constexpr bool can_merge = false;
	static constexpr int regs = 0;
	static constexpr int prefetch_slots = 1;
	float *d;    // [cap] ascending
	uint32_t *s; // [cap] bit 31 = "already expanded"
	int size;    // wave-uniform
	int limit;   // wave-uniform, <= cap
	int cursor;  // every entry below `cursor` is expanded

	__device__ __forceinline__ void bind(float *dd, uint32_t *ss) {
		d = dd, s = ss;
	}
	__device__ __forceinline__ void reset(int lim) {
		limit = lim, size = 0, cursor = 0;
	}
	// number of entries of [from, size) with distance < nd (the list is sorted, so they form a prefix of that range)
	__device__ __forceinline__ int lower_bound(float nd, int from) const {
		const int lane = lane_id();
		int p = from;
		for (int base = from; base < size; base += 64) {
			const int i = base + lane;
			const int c = __popcll(__ballot(i < size && d[i] < nd));
			p += c;
			if (c < 64)
				break;
		}
		return p;
	}
	// move entries [p, end) one cell up (end < cap), highest tile first so that nothing is overwritten before it is read
	__device__ __forceinline__ void shift_up(int p, int end) {
		const int lane = lane_id();
		for (int hi = end; hi > p;) {
			int lo = (hi - 1) & ~63;
			lo = lo < p ? p : lo;
			const int i = lo + lane;
			float vd = 0.f;
			uint32_t vs = 0;
			if (i < hi) {
				vd = d[i];
				vs = s[i];
			}
			wave_sync(); // every lane has its value before any lane stores
			if (i < hi) {
				d[i + 1] = vd;
				s[i + 1] = vs;
			}
			wave_sync();
			hi = lo;
		}
	}
	template <bool SKIP = false> // (WaveList's flavour switch; nothing to choose here)
	__device__ __forceinline__ bool insert(float nd, uint32_t ns) {
		const int p = lower_bound(nd, 0);
		if (p == limit)
			return false;
		shift_up(p, size < limit ? size : limit - 1);
		if (lane_id() == 0) {
			d[p] = nd;
			s[p] = ns;
		}
		wave_sync();
		if (size < limit)
			size++;
		if (p < cursor)
			cursor = p;
		return true;
	}
	__device__ __forceinline__ void get(int pos, float &od, uint32_t &os) const {
		od = d[pos]; // same address in every lane: one broadcast read
		os = s[pos];
	}
	__device__ __forceinline__ float last_distance() const {
		return d[size - 1];
	}
	__device__ __forceinline__ int first_unexpanded() {
		const int lane = lane_id();
		for (int base = cursor; base < size; base += 64) {
			const int i = base + lane;
			const unsigned long long m = __ballot(i < size && !(s[i] & EXPANDED_BIT));
			if (m) {
				cursor = base + __builtin_ctzll(m);
				return cursor;
			}
		}
		cursor = size;
		return -1;
	}
	__device__ __forceinline__ int first_unexpanded_entry(float &od, uint32_t &os) {
		const int pos = first_unexpanded();
		if (pos >= 0)
			get(pos, od, os);
		return pos;
	}
	// (WaveList::accept for the list in memory: the same sequence of inserts)
	__device__ __forceinline__ void accept(float cd, uint32_t cs, unsigned long long pass, float &radius) {
		while (pass) {
			const int j = __builtin_ctzll(pass);
			pass &= pass - 1;
			const float dj = read_lane(cd, j);
			if (size < limit || dj < radius) {
				insert(dj, read_lane(cs, j));
				radius = last_distance();
			}
		}
	}
	__device__ __forceinline__ void mark_expanded(int pos) {
		if (lane_id() == 0)
			s[pos] |= EXPANDED_BIT;
		wave_sync();
	}
};

// The candidates of a search that must tell admitted from rejected rows (tombstones, predicate): every accepted
// candidate waits here until it is expanded; expanded ones are of no further use (the result lives in the `top` list), so
// the queue is a sorted array with a moving head.  pop order = ascending distance, later arrivals before equal ones —
// exactly the order in which the single list with "expanded" marks hands them out.
struct CandQueue {
	float *d;
	uint32_t *s;
	int head, size, cap; // live entries: [head, size)

	__device__ __forceinline__ void bind(float *dd, uint32_t *ss, int capacity) {
		d = dd, s = ss, cap = capacity;
		head = size = 0;
	}
	__device__ __forceinline__ bool empty() const {
		return head == size;
	}
	__device__ __forceinline__ void front(float &od, uint32_t &os) const {
		od = d[head];
		os = s[head];
	}
	__device__ __forceinline__ void pop() {
		head++;
	}
	__device__ __forceinline__ void restart() {
		head = size = 0;
	}
	// false = out of space (the host re-runs the query with a larger queue)
	__device__ __forceinline__ bool push(float nd, uint32_t ns, float /*radius*/ = 0.f, bool /*top_full*/ = false) {
		const int lane = lane_id();
		if (size == cap) {
			if (head == 0)
				return false;
			const int live = size - head; // slide the live part down to the start, lowest tile first
			for (int base = 0; base < live; base += 64) {
				const int i = base + lane;
				float vd = 0.f;
				uint32_t vs = 0;
				if (i < live) {
					vd = d[head + i];
					vs = s[head + i];
				}
				wave_sync();
				if (i < live) {
					d[i] = vd;
					s[i] = vs;
				}
				wave_sync();
			}
			head = 0;
			size = live;
		}
		int p = head;
		for (int base = head; base < size; base += 64) {
			const int i = base + lane;
			const int c = __popcll(__ballot(i < size && d[i] < nd));
			p += c;
			if (c < 64)
				break;
		}
		for (int hi = size; hi > p;) {
			int lo = (hi - 1) & ~63;
			lo = lo < p ? p : lo;
			const int i = lo + lane;
			float vd = 0.f;
			uint32_t vs = 0;
			if (i < hi) {
				vd = d[i];
				vs = s[i];
			}
			wave_sync();
			if (i < hi) {
				d[i + 1] = vd;
				s[i + 1] = vs;
			}
			wave_sync();
			hi = lo;
		}
		if (lane == 0) {
			d[p] = nd;
			s[p] = ns;
		}
		wave_sync();
		size++;
		return true;
	}
};

// The same queue in registers for the common case — a few tombstones, an unselective predicate: at most 64 * E pending
// candidates, entry 0 is the next one to expand (ascending distance, later arrivals before equal ones, as above).
// When it is full the entry that would fall off — the last one, or the new one if it is farther — may be forgotten only
// if the reference would never expand it: the result list is full (so the radius can only shrink from here on) and the
// entry lies beyond the radius; popping it would end the search, and so does running out of candidates.  Otherwise push()
// reports false and the host re-runs the query with the unbounded CandQueue.
template <int E>
struct RegQueue {
	WaveList<E> c;
	__device__ __forceinline__ void restart() {
		c.reset(64 * E);
	}
	__device__ __forceinline__ bool empty() const {
		return c.size == 0;
	}
	__device__ __forceinline__ void front(float &od, uint32_t &os) const {
		c.get(0, od, os);
	}
	__device__ __forceinline__ void pop() {
		c.remove_first();
	}
	__device__ __forceinline__ bool push(float nd, uint32_t ns, float radius, bool top_full) {
		if (c.size < c.limit) {
			c.insert(nd, ns);
			return true;
		}
		const float ld = c.last_distance();
		const bool drop_last = nd <= ld; // the new entry goes before equal ones: the last one falls off
		const float gone = drop_last ? ld : nd;
		if (!(top_full && gone > radius))
			return false;
		if (drop_last)
			c.insert(nd, ns);
		return true;
	}
};

// ------------------------------------------------------------------------------------------------------
// VisitedSet (LDS)
// ------------------------------------------------------------------------------------------------------
//
// The compact form (round 4; VSS_VISITED_COMPACT=0 turns it off): 16-bit cells — tag + displacement — still
// EXACT; arithmetic and proof sketch in visited_compact.h.  A key whose displacement does not fit raises the lane's `bad` flag
// (a local of the gather): the caller reports a visited-set overflow and the host re-runs the query with the 32-bit table.
// Twice the cells of the 32-bit form in the same bytes: the sets of searches with limits of 257-512 stay in LDS.
// A visited set in LDS is probed with DS instructions, explicitly.  Through its generic pointer the compiler emits FLAT loads
// (volatile: `sc0 sc1`) and FLAT atomics: twice a DS instruction's latency through the address aperture, and — they count on
// vmcnt AND lgkmcnt — every wait for one of them is an `s_waitcnt vmcnt(0)`, i.e. for whatever global load the walker has in
// flight as well (its look-ahead's list requests).  Found in the ISA at the end of round 6 (rounds 1-6 ran every probe that way;
// it also tainted round 6's "compare-and-swap first or read first" A/B of the 32-bit form: its reads were FLAT, its swaps were not).
typedef __attribute__((address_space(3))) uint32_t lds_cell_t;
// (generic -> LDS through the integer value: the low 32 bits of a generic LDS address are the LDS offset; an addrspacecast's
//  null check is lowered by ROCm 7.2's hipcc to an instruction its own verifier rejects)
__device__ __forceinline__ lds_cell_t *as_lds(const uint32_t *p) {
	return (lds_cell_t *)(uint32_t)(uintptr_t)p;
}
__device__ __forceinline__ bool in_lds(const void *p) { // does this generic pointer name LDS?  (a compare with the aperture base)
	return __builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void *)p);
}
__device__ __forceinline__ uint32_t lds_load(const uint32_t *p) { // ds_read_b32, never cached in a register across a loop
	return __hip_atomic_load(as_lds(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t lds_cas(uint32_t *p, uint32_t expected, uint32_t desired) { // ds_cmpst_rtn_b32: the old value
	__hip_atomic_compare_exchange_strong(as_lds(p), &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	return expected;
}

struct VisitedSet {
	uint32_t *table; // LDS (the compact form: always), or HBM for the plain form of a large search; capacity = mask + 1 (power of two)
	uint32_t mask;
	uint32_t shift; // 32 - log2(capacity)
	uint32_t count; // wave-uniform
	uint32_t limit; // wave-uniform: inserting beyond this reports overflow
	uint32_t compact; // wave-uniform: 0 = 32-bit cells (a slot each); else the compact form (visited_compact.h: log2 cells | key bits << 8)
	enum { INSERTED = 0, SEEN_BEFORE = 1, LOST_TO_TWIN = 2 }; // probe(): how a key that is present got there

	__device__ __forceinline__ void clear() {
		const uint32_t words = compact ? (mask >> 1) : mask; // (two 16-bit cells per word; EMPTY_SLOT = both empty)
		count = 0;
		if (in_lds(table)) { // (wave-uniform) 16 bytes per lane and DS store; at least 256 words, a power of two
			typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
			typedef __attribute__((address_space(3))) u32x4 lds_u32x4_t;
			lds_u32x4_t *t4 = (lds_u32x4_t *)(uint32_t)(uintptr_t)table;
			const u32x4 empty = {EMPTY_SLOT, EMPTY_SLOT, EMPTY_SLOT, EMPTY_SLOT};
			for (uint32_t i = lane_id(); i <= (words >> 2); i += 64)
				t4[i] = empty;
			lds_sync();
			return;
		}
		for (uint32_t i = lane_id(); i <= words; i += 64)
			table[i] = EMPTY_SLOT;
		wave_sync();
	}

	// ---- compact form -------------------------------------------------------------------------------------
	// (register-lean on purpose: the 8-register list's kernel has none to spare.  Per lane: the cell index and the cell's
	// wanted content, whose low D bits ARE the displacement — both simply count up along the probe sequence.)
	__device__ __forceinline__ void home_of(uint32_t key, uint32_t &cell, uint32_t &want) const {
		compact_visited::home_of(key, compact, cell, want);
	}
	__device__ __forceinline__ bool placed_too_far(uint32_t want) const { // displacement 2^D - 1 is "empty": never stored
		return compact_visited::placed_too_far(want, compact);
	}
	__device__ __forceinline__ bool contains16(uint32_t key) const {
		uint32_t c, want;
		for (home_of(key, c, want); !placed_too_far(want); c = (c + 1) & mask, ++want) {
			const uint32_t half = (lds_load(&table[c >> 1]) >> ((c & 1) << 4)) & 0xFFFFu;
			if (half == want)
				return true;
			if (half == 0xFFFFu)
				return false;
		}
		return false; // (never placed that far: the insertion would have raised its lane's `bad`)
	}
	// One flat loop per key: READ the word that holds the cell — "my key is there" and "the cell is taken (next cell)" need no
	// atomic — and compare-and-swap only into a cell that read empty (a failed one means the word changed: read it again).
	// Round 6 (tools/microbench/walker_ops, profiles/r06c_walker_ops*.txt): rounds 4-5 ran a compare-and-swap first, on the guess
	// "both cells of the word are empty", inside two nested loops — 2.1k cycles for 32 ids at 15 % fill against 0.8k for the
	// 32-bit form; reading first is the faster order for THIS form (1.9k; seven ids in ten are "seen before", and a neighbour's
	// half is rarely empty once the set fills), while the 32-bit form keeps its compare-and-swap first (0.8k against 1.1k).
	__device__ __forceinline__ bool test_and_set16(uint32_t key, uint32_t &bad) {
		uint32_t c, want;
		home_of(key, c, want);
		for (;;) {
			if (placed_too_far(want)) {
				bad = 1;
				return false;
			}
			const uint32_t sh = (c & 1) << 4;
			const uint32_t cur = lds_load(&table[c >> 1]);
			const uint32_t half = (cur >> sh) & 0xFFFFu;
			if (half == want)
				return true;
			if (half == 0xFFFFu) {
				if (lds_cas(&table[c >> 1], cur, (cur & ~(0xFFFFu << sh)) | (want << sh)) == cur)
					return false;
				continue; // the word changed under us (the other half, or a twin of another lane): look again
			}
			c = (c + 1) & mask, ++want;
		}
	}
	// (lanes with equal keys run this in lockstep — same cells, same words read by the same instruction; only the outcome of
	// the compare-and-swap tells them apart — so a cell READ empty and then found holding the key was filled by a twin lane)
	__device__ __forceinline__ int probe16(uint32_t key, uint32_t &bad) {
		uint32_t c, want;
		for (home_of(key, c, want); !placed_too_far(want); c = (c + 1) & mask, ++want) {
			const uint32_t sh = (c & 1) << 4;
			uint32_t cur = lds_load(&table[c >> 1]);
			for (;;) {
				const uint32_t half = (cur >> sh) & 0xFFFFu;
				if (half == want)
					return SEEN_BEFORE;
				if (half != 0xFFFFu)
					break;
				const uint32_t old = lds_cas(&table[c >> 1], cur, (cur & ~(0xFFFFu << sh)) | (want << sh));
				if (old == cur)
					return INSERTED;
				if (((old >> sh) & 0xFFFFu) == want)
					return LOST_TO_TWIN;
				cur = old; // the other half changed, or another key took the cell (seen at the top)
			}
		}
		bad = 1;
		return INSERTED;
	}

	// Round 6: a compact set that has outgrown its cells MOVES instead of costing the query its work so far (rounds 4-5: the
	// query was repeated from the top — by the host in a further launch, then by its own walker — and the launch ended with
	// exactly those queries, its heaviest).  Every key is recovered from its cell (compact_visited::key_of: the cells are
	// invertible) and re-inserted into `bigger`, a table of 2^log2 32-bit cells in HBM; the set is the plain form from then on.
	// Exact by construction: the same keys, and the plain table never forgets one.
	__device__ __forceinline__ void migrate(uint32_t *bigger, uint32_t log2);

	// membership without insertion (the engine's look-ahead probes a list it may never expand)
	__device__ __forceinline__ bool contains(uint32_t key) const {
		if (compact)
			return contains16(key);
		uint32_t h = (key * 2654435761u) >> shift;
		if (in_lds(table)) { // (wave-uniform)
			for (;;) {
				const uint32_t cur = lds_load(&table[h]);
				if (cur == EMPTY_SLOT)
					return false;
				if (cur == key)
					return true;
				h = (h + 1) & mask;
			}
		}
		for (;;) {
			const uint32_t cur = *(volatile uint32_t *)&table[h];
			if (cur == EMPTY_SLOT)
				return false;
			if (cur == key)
				return true;
			h = (h + 1) & mask;
		}
	}

	// growing_hash_set_gt::set — returns the PREVIOUS membership (true = was already visited).  For lanes with distinct
	// keys; chunks of a list that may repeat a key go through probe() / mark_first_visit().
	__device__ __forceinline__ bool test_and_set(uint32_t key) {
		uint32_t bad = 0; // (the first key of an empty table: always fits)
		return test_and_set(key, bad);
	}
	__device__ __forceinline__ bool test_and_set(uint32_t key, uint32_t &bad) {
		if (compact)
			return test_and_set16(key, bad);
		uint32_t h = (key * 2654435761u) >> shift;
		if (in_lds(table)) { // (wave-uniform) the set lives in LDS: DS instructions
			for (;;) {
#ifdef VSS_VISITED_READ_FIRST32 // (A/B builds: tools/microbench/walker_ops)
				uint32_t old = lds_load(&table[h]);
				if (old == key)
					return true;
				if (old == EMPTY_SLOT)
					old = lds_cas(&table[h], EMPTY_SLOT, key);
#else
				const uint32_t old = lds_cas(&table[h], EMPTY_SLOT, key);
#endif
				if (old == EMPTY_SLOT)
					return false;
				if (old == key)
					return true;
				h = (h + 1) & mask;
			}
		}
		for (;;) {
			uint32_t old = atomicCAS(&table[h], EMPTY_SLOT, key);
			if (old == EMPTY_SLOT)
				return false;
			if (old == key)
				return true;
			h = (h + 1) & mask;
		}
	}

	// The same for all active lanes at once, telling apart how a key that is present got there: lanes holding equal
	// keys walk the same cells in lockstep, so a lane that READ an empty cell and then found its own key in it lost the
	// compare-and-swap to a twin lane of this very instruction.
	__device__ __forceinline__ int probe(uint32_t key, uint32_t &bad) {
		if (compact)
			return probe16(key, bad);
		uint32_t h = (key * 2654435761u) >> shift;
		if (in_lds(table)) { // (wave-uniform)
			for (;;) {
				const uint32_t cur = lds_load(&table[h]);
				if (cur == key)
					return SEEN_BEFORE;
				if (cur == EMPTY_SLOT) {
					const uint32_t old = lds_cas(&table[h], EMPTY_SLOT, key);
					if (old == EMPTY_SLOT)
						return INSERTED;
					if (old == key)
						return LOST_TO_TWIN;
				}
				h = (h + 1) & mask;
			}
		}
		for (;;) {
			const uint32_t cur = *(volatile uint32_t *)&table[h];
			if (cur == key)
				return SEEN_BEFORE;
			if (cur == EMPTY_SLOT) {
				const uint32_t old = atomicCAS(&table[h], EMPTY_SLOT, key);
				if (old == EMPTY_SLOT)
					return INSERTED;
				if (old == key)
					return LOST_TO_TWIN;
			}
			h = (h + 1) & mask;
		}
	}
};

// (a function of its own, called on the rare path: inlined into every gather of the 8-register list's kernel it cost that kernel
//  its scratch-free register allocation; everything travels by value, so no struct has to live in memory for the call)
__device__ __noinline__ void visited_move_cells(const uint32_t *old_table, uint32_t form, uint32_t *bigger, uint32_t log2) {
	VisitedSet nv;
	nv.table = bigger, nv.mask = (1u << log2) - 1, nv.shift = 32 - log2, nv.limit = 1u << 30, nv.count = 0, nv.compact = 0;
	for (uint32_t i = lane_id(); i <= nv.mask; i += 64)
		bigger[i] = EMPTY_SLOT;
	wave_sync();
	const uint32_t cells = 1u << compact_visited::cells_log2_of(form);
	for (uint32_t c = lane_id(); c < cells; c += 64) {
		const uint32_t half = (lds_load(&old_table[c >> 1]) >> ((c & 1) << 4)) & 0xFFFFu;
		if (half != 0xFFFFu)
			nv.test_and_set(compact_visited::key_of(c, half, form));
	}
	wave_sync();
}
__device__ __forceinline__ void VisitedSet::migrate(uint32_t *bigger, uint32_t log2) {
	visited_move_cells(table, compact, bigger, log2);
	table = bigger, mask = (1u << log2) - 1, shift = 32 - log2, limit = ((1u << log2) / 8) * 7, compact = 0;
}

// ------------------------------------------------------------------------------------------------------
// Distances
// ------------------------------------------------------------------------------------------------------
struct RowSpace {
	const float4 *vectors; // rows x V float4, zero padded
	uint32_t V;            // float4 chunks per row (= row stride)
	uint32_t G;            // lanes per row, power of two <= 64
	uint32_t logG;
	int metric; // 0 l2sq, 1 cosine, 2 ip (host-side dispatch key; kernels take it as the MT template parameter)
	uint32_t debug_rows; // debug builds (-DVSS_PARANOID): number of valid slots, and where to leave a note when an id is not one
	uint32_t *debug;
};

template <int MT>
__device__ __forceinline__ float finish_distance(float ab, float a2, float b2) {
	if (MT == 0)
		return ab;
	if (MT == 2)
		return 1.0f - ab;
	// metric_cos_gt, index_plugins.hpp:1021-1025
	if (a2 == 0.f && b2 == 0.f)
		return 0.f;
	if (a2 == 0.f || b2 == 0.f)
		return 1.f;
	return 1.0f - __fdiv_rn(ab, __fmul_rn(vss_sqrt(a2), vss_sqrt(b2)));
}

__device__ __forceinline__ float group_butterfly(float v, uint32_t G) {
	if (G > 32)
		v = __fadd_rn(v, lane_xor<32>(v));
	if (G > 16)
		v = __fadd_rn(v, lane_xor<16>(v));
	if (G > 8)
		v = __fadd_rn(v, lane_xor<8>(v));
	if (G > 4)
		v = __fadd_rn(v, lane_xor<4>(v));
	if (G > 2)
		v = __fadd_rn(v, lane_xor<2>(v));
	if (G > 1)
		v = __fadd_rn(v, lane_xor<1>(v));
	return v;
}

// Sum R per-lane partials over the 64 lanes with the butterfly's exact pairing (off = 32, 16, ..., 1) but a
// "transposing" schedule: at each of the first log2(R) steps a lane hands half of its rows to its partner, so R rows
// cost R-1 + log2(64/R) shuffles instead of 6R.  Afterwards v[0] of lane l is the total of row l / (64/R).
template <int OFF, int HALF, int R>
__device__ __forceinline__ void transpose_step(float (&v)[R]) {
	const bool upper = (lane_id() & OFF) != 0;
#pragma unroll
	for (int i = 0; i < HALF; ++i) {
		const float send = upper ? v[i] : v[i + HALF];
		const float keep = upper ? v[i + HALF] : v[i];
		v[i] = __fadd_rn(keep, lane_xor<OFF>(send));
	}
}
template <int R>
__device__ __forceinline__ void transposed_reduce64(float (&v)[R]) {
	static_assert(R == 16 || R == 8 || R == 4 || R == 2 || R == 1, "rows in flight");
	if constexpr (R == 16) { // (the solo search kernel: every row of an expansion in flight at once)
		transpose_step<32, 8>(v);
		transpose_step<16, 4>(v);
		transpose_step<8, 2>(v);
		transpose_step<4, 1>(v);
		v[0] = __fadd_rn(v[0], lane_xor<2>(v[0]));
		v[0] = __fadd_rn(v[0], lane_xor<1>(v[0]));
		return;
	} else if constexpr (R == 8) {
		transpose_step<32, 4>(v);
		transpose_step<16, 2>(v);
		transpose_step<8, 1>(v);
	} else if constexpr (R == 4) {
		transpose_step<32, 2>(v);
		transpose_step<16, 1>(v);
		v[0] = __fadd_rn(v[0], lane_xor<8>(v[0]));
	} else if constexpr (R == 2) {
		transpose_step<32, 1>(v);
		v[0] = __fadd_rn(v[0], lane_xor<16>(v[0]));
		v[0] = __fadd_rn(v[0], lane_xor<8>(v[0]));
	} else {
		v[0] = __fadd_rn(v[0], lane_xor<32>(v[0]));
		v[0] = __fadd_rn(v[0], lane_xor<16>(v[0]));
		v[0] = __fadd_rn(v[0], lane_xor<8>(v[0]));
	}
	v[0] = __fadd_rn(v[0], lane_xor<4>(v[0]));
	v[0] = __fadd_rn(v[0], lane_xor<2>(v[0]));
	v[0] = __fadd_rn(v[0], lane_xor<1>(v[0]));
}

// generic groups (G < 64).  The butterfly's pairing (off = G/2 ... 1) is fixed by the summation-order contract; how the
// partials travel is not.  Round 3: the offsets are compile-time (round 2 dispatched every single shuffle through a
// run-time switch: 5 x R branch ladders per pass at dimension 128 — the dominant cost of a scoring pass with a wide row
// window), and when the R rows of a pass fit the group (R <= G) they are reduced with the same transposing schedule as
// full-wave rows: at each of the first log2(R) steps a lane hands half of its rows to its partner, so R rows cost
// R - 1 + log2(G / R) shuffles instead of R x log2(G).  Afterwards v[0] of lane g (within its group) is the total of row
// slot g / (G / R).
template <int OFF, int R>
__device__ __forceinline__ void butterfly_step(float (&v)[R]) {
#pragma unroll
	for (int r = 0; r < R; ++r)
		v[r] = __fadd_rn(v[r], lane_xor<OFF>(v[r]));
}
template <int R>
__device__ __forceinline__ void group_reduce(float (&v)[R], uint32_t G) {
	if (G > 32)
		butterfly_step<32>(v);
	if (G > 16)
		butterfly_step<16>(v);
	if (G > 8)
		butterfly_step<8>(v);
	if (G > 4)
		butterfly_step<4>(v);
	if (G > 2)
		butterfly_step<2>(v);
	if (G > 1)
		butterfly_step<1>(v);
}
template <int G, int R> // powers of two, R <= G <= 32
__device__ __forceinline__ void group_reduce_transposed(float (&v)[R]) {
	static_assert(R <= G && G <= 32, "rows per pass must fit the lane group");
	if constexpr (R >= 2)
		transpose_step<G / 2, R / 2>(v);
	if constexpr (R >= 4)
		transpose_step<G / 4, R / 4>(v);
	if constexpr (R >= 8)
		transpose_step<G / 8, R / 8>(v);
	if constexpr (R >= 16)
		transpose_step<G / 16, R / 16>(v);
	constexpr int REST = G / R; // lanes still holding partials of the same row: offsets REST/2 ... 1 remain
	if constexpr (REST > 16)
		v[0] = __fadd_rn(v[0], lane_xor<16>(v[0]));
	if constexpr (REST > 8)
		v[0] = __fadd_rn(v[0], lane_xor<8>(v[0]));
	if constexpr (REST > 4)
		v[0] = __fadd_rn(v[0], lane_xor<4>(v[0]));
	if constexpr (REST > 2)
		v[0] = __fadd_rn(v[0], lane_xor<2>(v[0]));
	if constexpr (REST > 1)
		v[0] = __fadd_rn(v[0], lane_xor<1>(v[0]));
}
// true: v[0] holds the total of row slot g / (G / R) (transposed); false: plain butterflies ran, v[r] is row slot r's total
template <int R>
__device__ __forceinline__ bool group_reduce_rows(float (&v)[R], uint32_t G) {
#ifndef VSS_PLAIN_GROUP_REDUCE
	if constexpr (R >= 2 && R <= 32) {
		if (G == 32) {
			group_reduce_transposed<32, R>(v);
			return true;
		}
		if constexpr (R <= 16) {
			if (G == 16) {
				group_reduce_transposed<16, R>(v);
				return true;
			}
		}
		if constexpr (R <= 8) {
			if (G == 8) {
				group_reduce_transposed<8, R>(v);
				return true;
			}
		}
	}
#endif
	group_reduce<R>(v, G);
	return false;
}

template <int MT>
__device__ __forceinline__ void accumulate4(const float4 &q, const float4 &x, float &ab, float &b2) {
	if (MT == 0) {
		float t;
		t = __fsub_rn(q.x, x.x), ab = __fmaf_rn(t, t, ab);
		t = __fsub_rn(q.y, x.y), ab = __fmaf_rn(t, t, ab);
		t = __fsub_rn(q.z, x.z), ab = __fmaf_rn(t, t, ab);
		t = __fsub_rn(q.w, x.w), ab = __fmaf_rn(t, t, ab);
	} else if (MT == 1) {
		ab = __fmaf_rn(q.x, x.x, ab), b2 = __fmaf_rn(x.x, x.x, b2);
		ab = __fmaf_rn(q.y, x.y, ab), b2 = __fmaf_rn(x.y, x.y, b2);
		ab = __fmaf_rn(q.z, x.z, ab), b2 = __fmaf_rn(x.z, x.z, b2);
		ab = __fmaf_rn(q.w, x.w, ab), b2 = __fmaf_rn(x.w, x.w, b2);
	} else {
		ab = __fmaf_rn(q.x, x.x, ab);
		ab = __fmaf_rn(q.y, x.y, ab);
		ab = __fmaf_rn(q.z, x.z, ab);
		ab = __fmaf_rn(q.w, x.w, ab);
	}
}

// squared norm of the staged query in wave order (needed by cosine only); q_lds holds V float4 chunks
__device__ __forceinline__ float wave_query_norm(const RowSpace &sp, const float4 *q_lds) {
	const uint32_t g = lane_id() & (sp.G - 1);
	float a2 = 0.f;
	for (uint32_t c = g; c < sp.V; c += sp.G) {
		float4 q = q_lds[c];
		a2 = __fmaf_rn(q.x, q.x, a2);
		a2 = __fmaf_rn(q.y, q.y, a2);
		a2 = __fmaf_rn(q.z, q.z, a2);
		a2 = __fmaf_rn(q.w, q.w, a2);
	}
	return group_butterfly(a2, sp.G);
}

// out[j] = distance(query, row ids[j]) for j < n.  ids/out live in LDS.  NCH = chunks per lane known at compile
// time (V <= NCH * G), 0 = loop at run time.  R rows are in flight per lane group.
struct NoHook {
	static constexpr bool lds_only_sync = false;
	__device__ __forceinline__ void operator()() const {
	}
};
// does the hook type ask for an LDS-only sync at the end of wave_distances (its own loads — touches — stay in flight)?
template <class H, class = void>
struct hook_lds_only : std::false_type {};
template <class H>
struct hook_lds_only<H, std::void_t<decltype(H::lds_only_sync)>> : std::integral_constant<bool, H::lds_only_sync> {};
// `after_issue` (optional): called once, right after the row loads of the first pass have been issued.
template <int MT, int NCH, int R, class Hook = NoHook>
__device__ __forceinline__ void wave_distances(const RowSpace &sp, const float4 *q_lds, float qa2, const uint32_t *ids,
                                               int n_in, float *out, Hook after_issue = Hook()) {
	const int n = uniform(n_in);
	const uint32_t lane = lane_id();
	// More than one chunk per lane means V > 64: a full-wave group, known at COMPILE time — the chunk offsets of a row
	// (ch * G float4s) then become immediate offsets of the loads instead of 64-bit adds per load in the scoring waves'
	// prologue, which four waves per SIMD issue side by side before the first row load of the last of them goes out.
	const uint32_t G = NCH >= 2 ? 64u : sp.G, logG = NCH >= 2 ? 6u : sp.logG;
	const uint32_t g = lane & (G - 1);
	const uint32_t sub = lane >> logG;
	const int RG = 64 >> logG; // rows handled side by side by one register slot
	for (int base = 0; base < n; base += R * RG) {
		const float4 *row[R];
		int jraw[R];
#pragma unroll
		for (int r = 0; r < R; ++r) {
			jraw[r] = base + r * RG + (int)sub;
			const int j = jraw[r] < n ? jraw[r] : n - 1;
			row[r] = sp.vectors + (size_t)ids[j] * sp.V;
		}
		float ab[R], b2[R];
#pragma unroll
		for (int r = 0; r < R; ++r)
			ab[r] = 0.f, b2[r] = 0.f;
		// Loads are UNCONDITIONAL (row index clamped to n-1, chunk index always in range): a guarded load
		// (`ok ? row[c] : 0`) makes hipcc wait for each load inside its branch, i.e. 24 serialized HBM latencies.
		if constexpr (NCH > 0) {
			// instantiated only for V == NCH * G (the host falls back to the looping variant otherwise)
			float4 x[NCH][R];
#pragma unroll
			for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
				for (int r = 0; r < R; ++r)
					x[ch][r] = row[r][g + ch * G];
			// Wide row windows (the solo search kernel): every load is issued before the first FMA.  Left alone the machine
			// scheduler sinks the loads next to their uses to save registers — two rows, s_waitcnt vmcnt(0), two rows, ... —
			// i.e. R/2 serialized HBM round trips per pass (seen in the ISA; 12 us per expansion on the GPU).
			if constexpr (R >= 16)
				__builtin_amdgcn_sched_barrier(0);
			if constexpr (!std::is_same<Hook, NoHook>::value) {
				if (base == 0) {
					after_issue();
					__builtin_amdgcn_sched_barrier(0);
				}
			}
#pragma unroll
			for (int ch = 0; ch < NCH; ++ch) {
				const float4 q = q_lds[g + ch * G];
#pragma unroll
				for (int r = 0; r < R; ++r)
					accumulate4<MT>(q, x[ch][r], ab[r], b2[r]);
			}
		} else {
			// (the chunk loop is not wave-uniform — lanes beyond a short row sit it out — so the hook, which is wave-level
			// code, runs ahead of the loads here rather than in their shadow)
			if constexpr (!std::is_same<Hook, NoHook>::value) {
				if (base == 0)
					after_issue();
			}
			for (uint32_t c = g; c < sp.V; c += sp.G) {
				const float4 q = q_lds[c];
				float4 x[R];
#pragma unroll
				for (int r = 0; r < R; ++r)
					x[r] = row[r][c];
#pragma unroll
				for (int r = 0; r < R; ++r)
					accumulate4<MT>(q, x[r], ab[r], b2[r]);
			}
		}
		// (more than one chunk per lane means V > 64, i.e. a full-wave group: known at compile time for NCH >= 2)
		if (NCH >= 2 || G == 64) { // wave-uniform: one row per register slot
			transposed_reduce64<R>(ab);
			if (MT == 1)
				transposed_reduce64<R>(b2);
			constexpr int PER = 64 / R;
			const int j = base + (int)(lane / PER);
			if ((lane % PER) == 0 && j < n)
				out[j] = finish_distance<MT>(ab[0], qa2, b2[0]);
		} else {
			const bool transposed = group_reduce_rows<R>(ab, G);
			if (MT == 1)
				group_reduce_rows<R>(b2, G);
			if (transposed) { // wave-uniform: lane g holds row slot g / (G / R) in register 0
				const uint32_t per = G / R; // lanes per row slot
				const int j = base + (int)(g / per) * RG + (int)sub;
				if ((g & (per - 1)) == 0 && j < n)
					out[j] = finish_distance<MT>(ab[0], qa2, b2[0]);
			} else {
#pragma unroll
				for (int r = 0; r < R; ++r)
					if (g == 0 && jraw[r] < n)
						out[jraw[r]] = finish_distance<MT>(ab[r], qa2, b2[r]);
			}
		}
	}
	if constexpr (hook_lds_only<Hook>::value)
		lds_sync(); // (ids, query and distances are LDS; the hook's touch loads are nobody's business)
	else
		wave_sync();
}

// distance(query, one row), known to every lane without an LDS round trip: the n = 1 case of wave_distances
// (same chunk partition, same butterfly -> same bits)
template <int MT>
__device__ __forceinline__ float wave_distance_one(const RowSpace &sp, const float4 *q_lds, float qa2, uint32_t id) {
	const uint32_t g = lane_id() & (sp.G - 1);
	const float4 *row = sp.vectors + (size_t)id * sp.V;
	float ab = 0.f, b2 = 0.f;
	for (uint32_t c = g; c < sp.V; c += sp.G)
		accumulate4<MT>(q_lds[c], row[c], ab, b2);
	ab = group_butterfly(ab, sp.G);
	if (MT == 1)
		b2 = group_butterfly(b2, sp.G);
	return finish_distance<MT>(ab, qa2, b2);
}

// Stage one row of global memory (dim floats at `src`, not necessarily 16-byte aligned) as the query in LDS,
// zero padded to V float4 chunks.
__device__ __forceinline__ void stage_query(float4 *q_lds, const float *src, uint32_t dim, uint32_t V) {
	float *q = reinterpret_cast<float *>(q_lds);
	for (uint32_t i = lane_id(); i < V * 4; i += 64)
		q[i] = i < dim ? src[i] : 0.f;
	wave_sync();
}
// same, from a 16-byte aligned zero-padded row of the index (V float4 chunks)
__device__ __forceinline__ void stage_row(float4 *q_lds, const float4 *src, uint32_t V) {
	for (uint32_t i = lane_id(); i < V; i += 64)
		q_lds[i] = src[i];
	wave_sync();
}

} // namespace vss
