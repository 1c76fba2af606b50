// wave_primitives.h — per-wavefront building blocks of the HNSW kernels (gfx950, wave64).
//
// One wavefront (64 lanes) owns one query / one node being inserted / one neighbour list being repaired.
// Everything here is wave-synchronous: a workgroup is exactly one wave (__launch_bounds__(64)), LDS regions are
// private to it, and `wave_sync()` (an s_barrier of a one-wave group + LDS fence) orders cross-lane LDS traffic.
//
//   * WaveList      a sorted candidate list living in registers, entry p at (lane p%64, register p/64)
//                   — replaces usearch's `top` sorted_buffer_gt + `next` max_heap_gt (index.hpp:783-917, 620-773)
//   * VisitedSet    an exact open-addressing set in LDS — replaces growing_hash_set_gt (index.hpp:1018-1144)
//   * wave_distances  distances from one staged query to a handful of rows, rows read as coalesced float4
//                   streams, reduced with an xor-butterfly — replaces metric_punned_t (index_plugins.hpp:977-1053)
//
// Summation order ("wave order", restated on the CPU by oracle/hnsw_oracle.cpp dist_wave_order): a row of
// V = ceil(dim/4) float4 chunks is handled by G = min(64, pow2ceil(V)) lanes; lane g accumulates chunks
// g, g+G, g+2G, ... component by component with fmaf, then the G partial sums are combined by
// acc += shfl_xor(acc, off) for off = G/2 ... 1.  Compiled with -ffp-contract=off so nothing else fuses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vss {

constexpr uint32_t EMPTY_SLOT = 0xFFFFFFFFu;
constexpr uint32_t EXPANDED_BIT = 0x80000000u;
constexpr int LIST_REGS = 8; // 64 * 8 = 512 entries: the largest ef / ef_construction a register list can hold

__device__ __forceinline__ int lane_id() {
	return threadIdx.x & 63;
}
__device__ __forceinline__ void wave_sync() {
	__syncthreads();
}
__device__ __forceinline__ unsigned long long lanes_below(int lane) {
	return (1ull << lane) - 1ull;
}

// ------------------------------------------------------------------------------------------------------
// WaveList
// ------------------------------------------------------------------------------------------------------
struct WaveList {
	float d[LIST_REGS];
	uint32_t s[LIST_REGS]; // bit 31 = "already expanded"
	int size;              // wave-uniform
	int limit;             // wave-uniform capacity (<= 64 * nregs)
	int nregs;             // wave-uniform, registers in use

	__device__ __forceinline__ void reset(int lim) {
		limit = lim;
		nregs = (lim + 63) >> 6;
		size = 0;
#pragma unroll
		for (int r = 0; r < LIST_REGS; ++r) {
			d[r] = 0.f;
			s[r] = 0;
		}
	}

	// sorted_buffer_gt::insert(element, limit), index.hpp:880-891: position = lower_bound (the new element goes
	// BEFORE equal distances); rejected if it would land at `limit`; the last entry falls off when full.
	__device__ __forceinline__ bool insert(float nd, uint32_t ns) {
		const int lane = lane_id();
		int p = 0;
#pragma unroll
		for (int r = 0; r < LIST_REGS; ++r) {
			if (r < nregs) {
				bool lt = (r * 64 + lane < size) && (d[r] < nd);
				p += __popcll(__ballot(lt));
			}
		}
		if (p == limit)
			return false;
		float carry_d = 0.f;
		uint32_t carry_s = 0;
		const int src = (lane + 63) & 63;
#pragma unroll
		for (int r = 0; r < LIST_REGS; ++r) {
			if (r < nregs) {
				float rd = __shfl(d[r], src);
				uint32_t rs = __shfl(s[r], src);
				float in_d = lane == 0 ? carry_d : rd;
				uint32_t in_s = lane == 0 ? carry_s : rs;
				const int pos = r * 64 + lane;
				if (pos > p) {
					d[r] = in_d;
					s[r] = in_s;
				} else if (pos == p) {
					d[r] = nd;
					s[r] = ns;
				}
				carry_d = rd; // lane 0 of rd/rs holds the old lane-63 entry = carry into the next register
				carry_s = rs;
			}
		}
		if (size < limit)
			size++;
		return true;
	}

	__device__ __forceinline__ void get(int pos, float &od, uint32_t &os) const {
		od = 0.f;
		os = 0;
#pragma unroll
		for (int r = 0; r < LIST_REGS; ++r) {
			if (r == (pos >> 6)) {
				od = __shfl(d[r], pos & 63);
				os = __shfl(s[r], pos & 63);
			}
		}
	}

	__device__ __forceinline__ float last_distance() const {
		float od;
		uint32_t os;
		get(size - 1, od, os);
		return od;
	}

	__device__ __forceinline__ int first_unexpanded() const {
		const int lane = lane_id();
		int pos = -1;
#pragma unroll
		for (int r = 0; r < LIST_REGS; ++r) {
			if (r < nregs && pos < 0) {
				bool u = (r * 64 + lane < size) && !(s[r] & EXPANDED_BIT);
				unsigned long long m = __ballot(u);
				if (m)
					pos = r * 64 + __builtin_ctzll(m);
			}
		}
		return pos;
	}

	__device__ __forceinline__ void mark_expanded(int pos) {
		const int lane = lane_id();
#pragma unroll
		for (int r = 0; r < LIST_REGS; ++r)
			if (r == (pos >> 6) && lane == (pos & 63))
				s[r] |= EXPANDED_BIT;
	}

	// entry `pos` of this lane's registers (pos%64 must be this lane); used to dump the list
	__device__ __forceinline__ void dump(float *out_d, uint32_t *out_s) const {
		const int lane = lane_id();
#pragma unroll
		for (int r = 0; r < LIST_REGS; ++r) {
			if (r < nregs) {
				const int pos = r * 64 + lane;
				if (pos < size) {
					out_d[pos] = d[r];
					out_s[pos] = s[r] & ~EXPANDED_BIT;
				}
			}
		}
	}
};

// ------------------------------------------------------------------------------------------------------
// VisitedSet (LDS)
// ------------------------------------------------------------------------------------------------------
struct VisitedSet {
	uint32_t *table; // LDS, capacity = mask + 1 (power of two)
	uint32_t mask;
	uint32_t shift; // 32 - log2(capacity)
	uint32_t count; // wave-uniform
	uint32_t limit; // wave-uniform: inserting beyond this reports overflow

	__device__ __forceinline__ void clear() {
		for (uint32_t i = lane_id(); i <= mask; i += 64)
			table[i] = EMPTY_SLOT;
		count = 0;
		wave_sync();
	}

	// growing_hash_set_gt::set — returns the PREVIOUS membership (true = was already visited).
	// All active lanes may call it concurrently with distinct or equal keys.
	__device__ __forceinline__ bool test_and_set(uint32_t key) {
		uint32_t h = (key * 2654435761u) >> shift;
		for (;;) {
			uint32_t old = atomicCAS(&table[h], EMPTY_SLOT, key);
			if (old == EMPTY_SLOT)
				return false;
			if (old == key)
				return true;
			h = (h + 1) & mask;
		}
	}
};

// ------------------------------------------------------------------------------------------------------
// Distances
// ------------------------------------------------------------------------------------------------------
struct RowSpace {
	const float4 *vectors; // rows x V float4, zero padded
	uint32_t V;            // float4 chunks per row (= row stride)
	uint32_t G;            // lanes per row, power of two <= 64
	uint32_t logG;
	int metric; // 0 l2sq, 1 cosine, 2 ip
};

__device__ __forceinline__ float finish_distance(int metric, float ab, float a2, float b2) {
	if (metric == 0)
		return ab;
	if (metric == 2)
		return 1.0f - ab;
	// metric_cos_gt, index_plugins.hpp:1021-1025
	if (a2 == 0.f && b2 == 0.f)
		return 0.f;
	if (a2 == 0.f || b2 == 0.f)
		return 1.f;
	return 1.0f - __fdiv_rn(ab, __fmul_rn(__fsqrt_rn(a2), __fsqrt_rn(b2)));
}

__device__ __forceinline__ float group_butterfly(float v, uint32_t G) {
	for (uint32_t off = G >> 1; off >= 1; off >>= 1)
		v = __fadd_rn(v, __shfl_xor(v, off));
	return v;
}

__device__ __forceinline__ void accumulate4(int metric, const float4 &q, const float4 &x, float &ab, float &b2) {
	if (metric == 0) {
		float t;
		t = __fsub_rn(q.x, x.x), ab = __fmaf_rn(t, t, ab);
		t = __fsub_rn(q.y, x.y), ab = __fmaf_rn(t, t, ab);
		t = __fsub_rn(q.z, x.z), ab = __fmaf_rn(t, t, ab);
		t = __fsub_rn(q.w, x.w), ab = __fmaf_rn(t, t, ab);
	} else if (metric == 1) {
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
template <int NCH, int R>
__device__ __forceinline__ void wave_distances(const RowSpace &sp, const float4 *q_lds, float qa2, const uint32_t *ids,
                                               int n, float *out) {
	const uint32_t lane = lane_id();
	const uint32_t g = lane & (sp.G - 1);
	const uint32_t sub = lane >> sp.logG;
	const int RG = 64 >> sp.logG; // rows handled side by side by one register slot
	for (int base = 0; base < n; base += R * RG) {
		const float4 *row[R];
		int jraw[R];
#pragma unroll
		for (int r = 0; r < R; ++r) {
			jraw[r] = base + r * RG + (int)sub;
			int j = jraw[r] < n ? jraw[r] : n - 1;
			row[r] = sp.vectors + (size_t)ids[j] * sp.V;
		}
		float ab[R], b2[R];
#pragma unroll
		for (int r = 0; r < R; ++r)
			ab[r] = 0.f, b2[r] = 0.f;
		if (NCH > 0) {
			float4 x[NCH > 0 ? NCH : 1][R];
#pragma unroll
			for (int ch = 0; ch < NCH; ++ch) {
				const uint32_t c = g + ch * sp.G;
#pragma unroll
				for (int r = 0; r < R; ++r) {
					if (c < sp.V && base + r * RG < n)
						x[ch][r] = row[r][c];
					else
						x[ch][r] = make_float4(0.f, 0.f, 0.f, 0.f);
				}
			}
#pragma unroll
			for (int ch = 0; ch < NCH; ++ch) {
				const uint32_t c = g + ch * sp.G;
				if (c < sp.V) {
					const float4 q = q_lds[c];
#pragma unroll
					for (int r = 0; r < R; ++r)
						accumulate4(sp.metric, q, x[ch][r], ab[r], b2[r]);
				}
			}
		} else {
			for (uint32_t c = g; c < sp.V; c += sp.G) {
				const float4 q = q_lds[c];
				float4 x[R];
#pragma unroll
				for (int r = 0; r < R; ++r) {
					if (base + r * RG < n)
						x[r] = row[r][c];
					else
						x[r] = make_float4(0.f, 0.f, 0.f, 0.f);
				}
#pragma unroll
				for (int r = 0; r < R; ++r)
					accumulate4(sp.metric, q, x[r], ab[r], b2[r]);
			}
		}
#pragma unroll
		for (int r = 0; r < R; ++r) {
			if (base + r * RG < n) { // wave-uniform
				float s_ab = group_butterfly(ab[r], sp.G);
				float s_b2 = sp.metric == 1 ? group_butterfly(b2[r], sp.G) : 0.f;
				if (g == 0 && jraw[r] < n)
					out[jraw[r]] = finish_distance(sp.metric, s_ab, qa2, s_b2);
			}
		}
	}
	wave_sync();
}

// Stage one row of global memory (dim floats at `src`, not necessarily 16-byte aligned) as the query in LDS,
// zero padded to V float4 chunks.
__device__ __forceinline__ void stage_query(float4 *q_lds, const float *src, uint32_t dim, uint32_t V) {
	float *q = reinterpret_cast<float *>(q_lds);
	for (uint32_t i = lane_id(); i < V * 4; i += 64)
		q[i] = i < dim ? src[i] : 0.f;
	wave_sync();
}

} // namespace vss
