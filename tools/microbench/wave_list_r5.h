// wave_list_r5.h — the candidate list of rounds 1-5 (entry p at lane p % 64, register p / 64), frozen here for the A/B of
// tools/microbench/walker_ops.hip against round 6's blocked, right-aligned WaveList (csrc/wave_primitives.h).  Not product code.
#pragma once
#include "wave_primitives.h"
namespace vss {
template <int E>
struct WaveListR5 {
	static constexpr bool can_merge = true;
	static constexpr int regs = E;
	static constexpr int prefetch_slots = E <= 4 ? 2 : 1; // neighbour lists kept in flight (ListCache)
	float d[E];
	uint32_t s[E]; // bit 31 = "already expanded"
	int size;      // wave-uniform
	int limit;     // wave-uniform capacity (<= 64 * E)

	__device__ __forceinline__ void reset(int lim) {
		limit = uniform(lim); // (readfirstlane: tells the compiler these live in scalar registers — otherwise it keeps the
		size = 0;             //  list's bookkeeping in vector registers and branches on it through the exec mask)
#pragma unroll
		for (int r = 0; r < E; ++r) {
			d[r] = 0.f;
			s[r] = 0;
		}
	}

	// sorted_buffer_gt::insert(element, limit), index.hpp:880-891: position = lower_bound (the new element goes
	// BEFORE equal distances); rejected if it would land at `limit`; the last entry falls off when full.
	// SKIP (round 5): only the registers between the one that holds the insertion point and the one that holds the new last
	// entry change; the others are skipped behind wave-uniform branches.  The BUILD's walker uses it (ten fewer registers in the
	// 8-register list's phase-A kernel; rows/s unchanged within the box-to-box spread: profiles/r05k_build_*).  For the search
	// engine's walker at limits of 257-512 the straight-line form — eight independent shift chains the hardware overlaps — is
	// the faster one (accept phase 4.5k against 5.5k ticks per expansion with the branches,
	// profiles/r05k_skipping_insert_in_the_search_walker_slower_*), so searches keep it.
	template <bool SKIP = false>
	__device__ __forceinline__ bool insert(float nd, uint32_t ns) {
		const int lane = lane_id();
		int p = 0;
#pragma unroll
		for (int r = 0; r < E; ++r) {
			const bool lt = (r * 64 + lane < size) && (d[r] < nd);
			p += __popcll(__ballot(lt));
		}
		if (p == limit)
			return false;
		// entries p .. size - 1 move one position up (the one that would land at `limit` falls off)
		const int first_r = p >> 6, last_r = (size < limit ? size : limit - 1) >> 6;
		float carry_d = 0.f;
		uint32_t carry_s = 0;
#pragma unroll
		for (int r = 0; r < E; ++r) {
			if (!SKIP || E <= 2 || (r >= first_r && r <= last_r)) {
				const float in_d = shift_up_one(carry_d, d[r]);
				const uint32_t in_s = shift_up_one(carry_s, s[r]);
				if (r + 1 < E) { // the entry leaving this register enters lane 0 of the next one
					carry_d = read_lane(d[r], 63);
					carry_s = read_lane(s[r], 63);
				}
				const int pos = r * 64 + lane;
				d[r] = pos > p ? in_d : (pos == p ? nd : d[r]);
				s[r] = pos > p ? in_s : (pos == p ? ns : s[r]);
			}
		}
		size = uniform(size < limit ? size + 1 : size);
		return true;
	}

	// Insert up to 64 elements at once — one per lane, those flagged in `take` — with the result the sequential inserts
	// (in lane order, each evicting the last entry once the list is full) would leave, PROVIDED no two distances involved
	// are equal: then the outcome is the `limit` smallest of (list U candidates) whatever the order.  With a tie (or a NaN)
	// the order matters, nothing is changed and false is returned: the caller inserts one by one.
	// stage_d / stage_s: LDS scratch of at least `limit` cells owned by this wave.
	__device__ __forceinline__ bool merge(float cd, uint32_t cs, unsigned long long take, float *stage_d, uint32_t *stage_s) {
		const int lane = lane_id();
		const bool mine = (take >> lane) & 1ull;
		int rank = 0, base = 0;
		bool tie = mine && !(cd == cd);
		int shift[E];
#pragma unroll
		for (int r = 0; r < E; ++r)
			shift[r] = 0;
		for (unsigned long long rest = take; rest; rest &= rest - 1) {
			const int j = __builtin_ctzll(rest);
			const float dj = read_lane(cd, j);
			rank += (mine && dj < cd) ? 1 : 0;
			tie = tie || (mine && dj == cd && j != lane);
			int below = 0;
#pragma unroll
			for (int r = 0; r < E; ++r) {
				const bool valid = r * 64 + lane < size;
				below += __popcll(__ballot(valid && d[r] < dj));
				shift[r] += (valid && dj < d[r]) ? 1 : 0;
				tie = tie || (valid && dj == d[r]);
			}
			if (lane == j)
				base = below;
		}
		if (__ballot(tie))
			return false;
#pragma unroll
		for (int r = 0; r < E; ++r) {
			const int pos = r * 64 + lane;
			const int np = pos + shift[r];
			if (pos < size && np < limit) {
				stage_d[np] = d[r];
				stage_s[np] = s[r];
			}
		}
		if (mine && base + rank < limit) {
			stage_d[base + rank] = cd;
			stage_s[base + rank] = cs;
		}
		lds_sync(); // (the staging rows are LDS: global loads issued ahead of time stay in flight)
		const int grown = size + __popcll(take);
		size = uniform(grown < limit ? grown : limit);
#pragma unroll
		for (int r = 0; r < E; ++r) {
			const int pos = r * 64 + lane;
			if (pos < size) {
				d[r] = stage_d[pos];
				s[r] = stage_s[pos];
			}
		}
		lds_sync();
		return true;
	}

	__device__ __forceinline__ void get(int pos, float &od, uint32_t &os) const {
		od = 0.f;
		os = 0;
		pos = uniform(pos);
#pragma unroll
		for (int r = 0; r < E; ++r) {
			if (E == 1 || r == (pos >> 6)) {
				od = read_lane(d[r], pos & 63);
				os = read_lane(s[r], pos & 63);
			}
		}
	}

	__device__ __forceinline__ float last_distance() const {
		float od;
		uint32_t os;
		get(size - 1, od, os);
		return od;
	}

	// does some entry carry exactly this distance?  (the pipelined level search: an exact tie decides an order by position,
	// and is left to the one-by-one path)
	__device__ __forceinline__ bool holds_distance(float x) const {
		const int lane = lane_id();
		bool any = false;
#pragma unroll
		for (int r = 0; r < E; ++r)
			any = any || ((r * 64 + lane < size) && d[r] == x);
		return __ballot(any) != 0ull;
	}

	__device__ __forceinline__ int first_unexpanded() const {
		const int lane = lane_id();
		int pos = -1;
#pragma unroll
		for (int r = 0; r < E; ++r) {
			if (pos < 0) {
				const bool u = (r * 64 + lane < size) && !(s[r] & EXPANDED_BIT);
				const unsigned long long m = __ballot(u);
				if (m)
					pos = r * 64 + __builtin_ctzll(m);
			}
		}
		return pos;
	}

	// the first unexpanded entry behind position `pos` (-1: none)
	__device__ __forceinline__ int next_unexpanded(int pos) const {
		const int lane = lane_id();
		int found = -1;
#pragma unroll
		for (int r = 0; r < E; ++r) {
			if (found < 0) {
				const int p = r * 64 + lane;
				const bool u = p > pos && p < size && !(s[r] & EXPANDED_BIT);
				const unsigned long long m = __ballot(u);
				if (m)
					found = r * 64 + __builtin_ctzll(m);
			}
		}
		return found;
	}

	__device__ __forceinline__ void mark_expanded(int pos) {
		const int lane = lane_id();
#pragma unroll
		for (int r = 0; r < E; ++r)
			if (r * 64 + lane == pos)
				s[r] |= EXPANDED_BIT;
	}

	// drop entry 0: every entry moves one position down
	__device__ __forceinline__ void remove_first() {
#pragma unroll
		for (int r = 0; r < E; ++r) {
			float in_d = 0.f;
			uint32_t in_s = 0;
			if (r + 1 < E) { // lane 0 of the next register enters lane 63 of this one
				in_d = read_lane(d[r + 1], 0);
				in_s = read_lane(s[r + 1], 0);
			}
			d[r] = shift_down_one(in_d, d[r]);
			s[r] = shift_down_one(in_s, s[r]);
		}
		size = uniform(size - 1);
	}

	// dump the list (ascending) into LDS arrays
	__device__ __forceinline__ void dump(float *out_d, uint32_t *out_s) const {
		const int lane = lane_id();
#pragma unroll
		for (int r = 0; r < E; ++r) {
			const int pos = r * 64 + lane;
			if (pos < size) {
				out_d[pos] = d[r];
				out_s[pos] = s[r] & ~EXPANDED_BIT;
			}
		}
	}
};

} // namespace vss
