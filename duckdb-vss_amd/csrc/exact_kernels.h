// exact_kernels.h — brute-force (exact) top-k, the array_* scalar functions and the multi-shard merge.
//
//   k_row_norms        |x|^2 per stored row (one streaming pass, HBM-bound)
//   k_exact_scores     S[B x C] = f(Q . X^T) on MFMA f32 (v_mfma_f32_32x32x2_f32), LDS-staged 128x128x32 tiles.
//                      Replaces the distance loop of usearch search_exact_ (index.hpp:4004-4019) for a whole
//                      batch of queries at once; scores are RANKING scores (|x|^2 - 2 q.x, -q.x/|x|, -q.x).
//   k_exact_select     per query: fold one chunk of scores into a running top-K' (K' = k + slack)
//   k_exact_rerank     per query: recompute the K' survivors with the exact wave-order metric, sort, emit k
//   k_array_distance   array_distance / array_cosine_distance / array_negative_inner_product over a column
//   k_merge_topk       k-way merge of per-shard results after the RCCL all-gather
#pragma once
#include "hnsw_kernels.h"

namespace vss {

// lexicographic (score, index) comparison
__device__ __forceinline__ bool lex_less(float s1, uint32_t i1, float s2, uint32_t i2) {
	return s1 < s2 || (s1 == s2 && i1 < i2);
}

#ifdef VSS_ENGINE_TU // plain kernels are defined once, in the engine's translation unit
// ---------------------------------------------------------------------------------------------------------
__global__ void k_row_norms(const float4 *vectors, uint32_t V, uint32_t G, uint32_t logG, uint32_t rows,
                            float *out) {
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t g = lane & (G - 1), sub = lane >> logG, RG = 64 >> logG;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
	for (uint32_t base = wave * RG; base < rows; base += n_waves * RG) {
		const uint32_t row = base + sub;
		float a2 = 0.f;
		if (row < rows) {
			const float4 *rp = vectors + (size_t)row * V;
			for (uint32_t c = g; c < V; c += G) {
				float4 x = rp[c];
				a2 = __fmaf_rn(x.x, x.x, a2);
				a2 = __fmaf_rn(x.y, x.y, a2);
				a2 = __fmaf_rn(x.z, x.z, a2);
				a2 = __fmaf_rn(x.w, x.w, a2);
			}
		}
		a2 = group_butterfly(a2, G);
		if (g == 0 && row < rows)
			out[row] = a2;
	}
}

// ---------------------------------------------------------------------------------------------------------
// MFMA score tile.  Block = 256 threads (4 waves as 2x2), block tile 128 queries x 128 rows, wave tile 64 x 64 =
// 2 x 2 MFMA 32x32 accumulators, K step 32.
// Round 2, measured on the GPU (profiles/README.md, r02b / r02d): what moved the kernel was occupancy — four workgroups
// per CU instead of three (below).  Measured and NOT kept, all within noise of or below that: K-major operands read with
// ds_read_b128 plus register-staged prefetch (two LDS buffers / one barrier, or one buffer), a K step of 16 with and
// without a second LDS buffer and the next step's loads in flight during the MFMAs, and software-pipelined operand reads
// inside the wave (sched_group_barrier: reads of step k+1 ahead of the MFMAs of step k).
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef VSS_EXACT_WAVES
#define VSS_EXACT_WAVES 4 // waves per workgroup: 4 (2 x 2, 64 x 64 each) or 8 (4 x 2, 32 x 64 each) — A/B builds
#endif
constexpr int XT_BM = 128, XT_BN = 128, XT_BK = 32, XT_LD = 129;
constexpr int XT_WAVES = VSS_EXACT_WAVES, XT_THREADS = 64 * XT_WAVES, XT_MI = 8 / XT_WAVES; // MFMA row tiles per wave
static_assert(XT_WAVES == 4 || XT_WAVES == 8, "score tile waves");

struct ExactArgs {
	const float4 *queries; // B x V float4 (zero padded)
	const float4 *vectors; // rows x V float4
	const float *row_norm2; // rows
	const float *query_norm2; // B
	const int64_t *keys;    // rows (tombstones are excluded)
	uint32_t V;
	uint32_t n_queries;
	uint32_t row_begin, row_end; // chunk of rows being scored
	uint32_t chunk_stride;       // leading dimension of `scores`
	int metric;
	float *scores; // n_queries x chunk_stride
	uint32_t probe; // diagnostics only (VSS_EXACT_PROBE): 1 = no score stores, 2 = no global loads after the prologue, 4 = no barrier;
	                // k_exact_scores_v3 / _v4 also: 8 = stagger the two workgroups of a compute unit, 16 = no LDS writes after the
	                // prologue (v3), 32 = no epilogue
	// Round 4, the select folded into the epilogue (k_exact_scores_v2 only; all NULL / 0 = store every score): once every
	// query's running top-K' is full, a score can only matter if it beats the K'-th best so far — (tau_s, tau_i)[q] =
	// (best_s, best_i)[q][KP - 1] as of the last select — so the epilogue appends just those survivors to the query's
	// candidate buffer (cand_cap cells; the count runs on past it, the select raises `overflow` and the host reruns the search
	// the plain way) and the 128 KiB of scores per query and chunk are neither written nor read back.
	const float *best_s;
	const uint32_t *best_i;
	uint32_t KP;
	uint32_t cand_cap;
	uint32_t *cand_cnt; // n_queries
	float *cand_s;      // n_queries x cand_cap
	uint32_t *cand_i;
};

// Register budget pinned to 4 waves per SIMD (128 registers, accumulators included): left alone the compiler takes 92
// VGPRs + 64 AGPRs = 3 waves per SIMD, i.e. three of the four workgroups the LDS admits per CU; with four, 0.73-0.745 of
// the f32 matrix peak against 0.65 (profiles/r02d_exact_ab.json).  -DVSS_EXACT_WAVES_PER_EU=n rebuilds it otherwise.
#ifndef VSS_EXACT_WAVES_PER_EU
#define VSS_EXACT_WAVES_PER_EU 4
#endif
#define VSS_EXACT_OCCUPANCY __attribute__((amdgpu_waves_per_eu(VSS_EXACT_WAVES_PER_EU, VSS_EXACT_WAVES_PER_EU)))
__global__ __launch_bounds__(XT_THREADS) VSS_EXACT_OCCUPANCY void k_exact_scores(ExactArgs a) {
	__shared__ float As[XT_BK * XT_LD];
	__shared__ float Bs[XT_BK * XT_LD];
	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int wm = wave >> 1, wn = wave & 1; // this wave's 32*XT_MI x 64 part of the tile
	const uint32_t q0 = blockIdx.y * XT_BM;
	const uint32_t r0 = a.row_begin + blockIdx.x * XT_BN;
	const uint32_t n_rows_total = a.row_end;

	f32x16 acc[XT_MI][2];
#pragma unroll
	for (int i = 0; i < XT_MI; ++i)
#pragma unroll
		for (int j = 0; j < 2; ++j)
#pragma unroll
			for (int e = 0; e < 16; ++e)
				acc[i][j][e] = 0.f;

	constexpr int F4 = XT_BK / 4;           // float4 chunks along K per step
	constexpr int ROWS_PER_PASS = XT_THREADS / F4; // rows one pass of the workgroup stages
	constexpr int PASSES = 128 / ROWS_PER_PASS;
	const int f = tid % F4; // which float4 along K
	const int rr = tid / F4;
	// the rows this thread stages (clamped: tiles at the edges re-read the last row; their results are never stored)
	const float4 *qsrc[PASSES], *xsrc[PASSES];
#pragma unroll
	for (int p = 0; p < PASSES; ++p) {
		uint32_t qi = q0 + rr + ROWS_PER_PASS * p;
		qi = qi < a.n_queries ? qi : a.n_queries - 1;
		uint32_t ri = r0 + rr + ROWS_PER_PASS * p;
		ri = ri < n_rows_total ? ri : n_rows_total - 1;
		qsrc[p] = a.queries + (size_t)qi * a.V;
		xsrc[p] = a.vectors + (size_t)ri * a.V;
	}
	// Loads are unconditional (a chunk index beyond the row is clamped and its values zeroed afterwards): behind a
	// branch hipcc waits for each load where it is issued, and the staging loads of a step would arrive one by one.
	auto load_step = [&](uint32_t k0, float4 (&qa)[PASSES], float4 (&xb)[PASSES]) {
		const uint32_t c = (k0 >> 2) + f;
		const uint32_t cc = c < a.V ? c : a.V - 1;
#pragma unroll
		for (int p = 0; p < PASSES; ++p) {
			qa[p] = qsrc[p][cc];
			xb[p] = xsrc[p][cc];
		}
		if (c >= a.V) {
#pragma unroll
			for (int p = 0; p < PASSES; ++p)
				qa[p] = xb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
		}
	};
	auto store_step = [&](float *A, float *B, const float4 (&qa)[PASSES], const float4 (&xb)[PASSES]) {
#pragma unroll
		for (int p = 0; p < PASSES; ++p) {
			const int row = rr + ROWS_PER_PASS * p;
			A[(4 * f + 0) * XT_LD + row] = qa[p].x;
			A[(4 * f + 1) * XT_LD + row] = qa[p].y;
			A[(4 * f + 2) * XT_LD + row] = qa[p].z;
			A[(4 * f + 3) * XT_LD + row] = qa[p].w;
			B[(4 * f + 0) * XT_LD + row] = xb[p].x;
			B[(4 * f + 1) * XT_LD + row] = xb[p].y;
			B[(4 * f + 2) * XT_LD + row] = xb[p].z;
			B[(4 * f + 3) * XT_LD + row] = xb[p].w;
		}
	};
	auto multiply_step = [&](const float *A, const float *B) {
		const float *ap = A + (lane >> 5) * XT_LD + wm * 32 * XT_MI + (lane & 31);
		const float *bp = B + (lane >> 5) * XT_LD + wn * 64 + (lane & 31);
#pragma unroll
		for (int kk = 0; kk < XT_BK; kk += 2) {
			float av[XT_MI], bv[2];
#pragma unroll
			for (int i = 0; i < XT_MI; ++i)
				av[i] = ap[kk * XT_LD + i * 32];
#pragma unroll
			for (int j = 0; j < 2; ++j)
				bv[j] = bp[kk * XT_LD + j * 32];
#pragma unroll
			for (int i = 0; i < XT_MI; ++i)
#pragma unroll
				for (int j = 0; j < 2; ++j)
					acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
		}
	};
	const uint32_t K = a.V * 4;
	for (uint32_t k0 = 0; k0 < K; k0 += XT_BK) {
		float4 qa[PASSES], xb[PASSES];
		load_step(k0, qa, xb);
		store_step(As, Bs, qa, xb);
		__syncthreads();
		multiply_step(As, Bs);
		__syncthreads();
	}

	// epilogue: C[row = (e&3) + 8*(e>>2) + 4*(lane>>5)][col = lane&31]
#pragma unroll
	for (int i = 0; i < XT_MI; ++i) {
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			const uint32_t col = r0 + wn * 64 + j * 32 + (lane & 31);
			const bool col_ok = col < n_rows_total;
			float xn2 = 0.f;
			bool live = false;
			if (col_ok) {
				xn2 = a.row_norm2[col];
				live = a.keys[col] != FREE_KEY;
			}
#pragma unroll
			for (int e = 0; e < 16; ++e) {
				const uint32_t qi = q0 + wm * 32 * XT_MI + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
				if (qi < a.n_queries && col - a.row_begin < a.chunk_stride) {
					const float dot = acc[i][j][e];
					float s;
					if (a.metric == 0)
						s = xn2 - 2.f * dot;
					else if (a.metric == 2)
						s = -dot;
					else
						s = xn2 > 0.f ? -dot * rsqrtf(xn2) : 0.f;
					if (!col_ok || !live)
						s = __builtin_inff();
					a.scores[(size_t)qi * a.chunk_stride + (col - a.row_begin)] = s;
				}
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// Round 3: the same score tile as a software pipeline.  What round 2's kernel does per K = 32 step — global loads, wait,
// 64 scalar transposing LDS writes, barrier, 64 scalar operand reads feeding 64 MFMAs, barrier — leaves the matrix pipe
// idle a quarter of the time: the four workgroups of a compute unit start together and stay in step, so their load /
// store / barrier phases coincide instead of filling each other's gaps (MFMA-busy 0.75 whatever the buffering).  Here a
// wave's MFMA stream never stops:
//   * two LDS buffers, ONE barrier per step; the global loads of step k+2 are issued during step k and parked in
//     registers, written to the idle buffer during step k+1's MFMAs;
//   * operands are stored as they arrive — row-major [row][k], 36-float stride, ds_write_b128 — and read with
//     ds_read_b128: a lane's four k values feed four consecutive MFMAs.  v_mfma_f32_32x32x2 takes its two k slices
//     from the two half-waves; which k a half-wave supplies is free as long as A and B agree (a dot product does not
//     care about the order of its terms), so half h of a group of eight k's simply owns k = 4h .. 4h+3: 16 wide LDS reads
//     per step instead of 64 narrow ones, no transposition anywhere;
//   * the operands of k-group g+1 are read while the 16 MFMAs of group g run.
// 73.7 KB of LDS per workgroup -> two workgroups (eight waves) per compute unit, 256 registers per lane available.
constexpr int X2_LD = 36; // floats per staged row: 32 + 4 (bank-conflict-free b128 reads and writes)
// TN = 32-row MFMA tiles of data rows per wave: 2 -> block tile 128 queries x 128 rows, 73.7 KB of LDS, two workgroups
// (eight waves) per compute unit; 4 -> 128 x 256, 110.6 KB, ONE workgroup per compute unit and one wave per SIMD with 128
// accumulator registers: a quarter fewer LDS reads and global loads per MFMA, and a barrier that four waves with a SIMD
// each reach in step.
template <int TN>
struct X2Shape {
	static constexpr int BN = 64 * TN;                  // data rows per block tile
	static constexpr int A_TILE = 128 * X2_LD;          // floats
	static constexpr int B_TILE = BN * X2_LD;
	static constexpr uint32_t LDS_BYTES = 2 * (A_TILE + B_TILE) * 4;
	static constexpr int WAVES_PER_EU = TN == 2 ? 2 : 1;
};

template <int TN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(X2Shape<TN>::WAVES_PER_EU, X2Shape<TN>::WAVES_PER_EU))) void
k_exact_scores_v2(ExactArgs a) {
	using S = X2Shape<TN>;
	constexpr int PB = S::BN / 32; // staging passes over the data rows (32 rows per pass)
	extern __shared__ __attribute__((aligned(16))) unsigned char x2_smem[];
	float *const lds = reinterpret_cast<float *>(x2_smem); // [buf][A | B][row][36]
	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int wm = wave >> 1, wn = wave & 1; // this wave's 64 x (32 TN) part of the block tile
	const uint32_t q0 = blockIdx.y * 128u;
	const uint32_t r0 = a.row_begin + blockIdx.x * (uint32_t)S::BN;
	const uint32_t n_rows_total = a.row_end;
	// a filtered pass that has overflowed is repeated the plain way by the host: its remaining launches have nothing to add.
	// (The word is written by k_exact_select only — launches ordered against this one on the pass's stream — so every thread of
	// this launch reads the same value: the exit is uniform, no wave leaves others at a barrier.)
	if (a.cand_cnt && __hip_atomic_load(&a.cand_cnt[a.n_queries], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
		return;
	// filtered epilogue: the thresholds of this tile's 128 queries (visible after the prologue's barrier)
	__shared__ float tau_s[128];
	__shared__ uint32_t tau_i[128];
	if (a.cand_cnt && tid < 128) {
		const uint32_t qi = q0 + tid < a.n_queries ? q0 + tid : a.n_queries - 1;
		tau_s[tid] = a.best_s[(size_t)qi * a.KP + a.KP - 1];
		tau_i[tid] = a.best_i[(size_t)qi * a.KP + a.KP - 1];
	}

	f32x16 acc[2][TN];
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int j = 0; j < TN; ++j)
#pragma unroll
			for (int e = 0; e < 16; ++e)
				acc[i][j][e] = 0.f;

	// staging: thread -> (float4 f of the step's eight, rows rr + 32 p); rows beyond the edge re-read the last one
	const int f = tid & 7, rr = tid >> 3;
	const float4 *qsrc[4], *xsrc[PB];
#pragma unroll
	for (int p = 0; p < 4; ++p) {
		uint32_t qi = q0 + rr + 32 * p;
		qi = qi < a.n_queries ? qi : a.n_queries - 1;
		qsrc[p] = a.queries + (size_t)qi * a.V;
	}
#pragma unroll
	for (int p = 0; p < PB; ++p) {
		uint32_t ri = r0 + rr + 32 * p;
		ri = ri < n_rows_total ? ri : n_rows_total - 1;
		xsrc[p] = a.vectors + (size_t)ri * a.V;
	}
	const uint32_t steps = (a.V + 7) / 8;
	float4 qa[4], xb[PB];
	auto load_step = [&](uint32_t step) { // unconditional loads (clamped chunk, zeroed afterwards), see k_exact_scores
		const uint32_t c = step * 8 + f;
		const uint32_t cc = c < a.V ? c : a.V - 1;
#pragma unroll
		for (int p = 0; p < 4; ++p)
			qa[p] = qsrc[p][cc];
#pragma unroll
		for (int p = 0; p < PB; ++p)
			xb[p] = xsrc[p][cc];
		if (c >= a.V) {
#pragma unroll
			for (int p = 0; p < 4; ++p)
				qa[p] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
			for (int p = 0; p < PB; ++p)
				xb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
		}
	};
	auto store_step = [&](int buf) {
		float *A = lds + buf * (S::A_TILE + S::B_TILE), *B = A + S::A_TILE;
#pragma unroll
		for (int p = 0; p < 4; ++p)
			*reinterpret_cast<float4 *>(A + (rr + 32 * p) * X2_LD + 4 * f) = qa[p];
#pragma unroll
		for (int p = 0; p < PB; ++p)
			*reinterpret_cast<float4 *>(B + (rr + 32 * p) * X2_LD + 4 * f) = xb[p];
	};
	// operand addresses of this lane: row (lane & 31) of each 32-row MFMA tile, k slice 4 * (lane >> 5) of every group of 8
	const int a_off = (wm * 64 + (lane & 31)) * X2_LD + 4 * (lane >> 5);
	const int b_off = (wn * 32 * TN + (lane & 31)) * X2_LD + 4 * (lane >> 5);
	auto read_group = [&](int buf, int g, float4 (&av)[2], float4 (&bv)[TN]) {
		const float *A = lds + buf * (S::A_TILE + S::B_TILE), *B = A + S::A_TILE;
#pragma unroll
		for (int i = 0; i < 2; ++i)
			av[i] = *reinterpret_cast<const float4 *>(A + a_off + i * 32 * X2_LD + g * 8);
#pragma unroll
		for (int j = 0; j < TN; ++j)
			bv[j] = *reinterpret_cast<const float4 *>(B + b_off + j * 32 * X2_LD + g * 8);
	};
	auto mfma_group = [&](const float4 (&av)[2], const float4 (&bv)[TN]) {
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
				acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
				acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
				acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
				acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
	};

	// prologue: step 0 into buffer 0, step 1 on its way
	load_step(0);
	store_step(0);
	if (steps > 1)
		load_step(1);
	__syncthreads();
	float4 av0[2], bv0[TN], av1[2], bv1[TN];
	read_group(0, 0, av0, bv0);
	for (uint32_t k = 0; k < steps; ++k) {
		const int buf = (int)(k & 1);
		read_group(buf, 1, av1, bv1);
		mfma_group(av0, bv0);
		read_group(buf, 2, av0, bv0);
		if (k + 1 < steps)
			store_step(buf ^ 1); // the registers of step k+1 (loaded a whole step ago) go to the idle buffer
		mfma_group(av1, bv1);
		read_group(buf, 3, av1, bv1);
		if (k + 2 < steps && !(a.probe & 2u))
			load_step(k + 2); // lands during the next step's MFMAs
		mfma_group(av0, bv0);
		// the barrier sits BEFORE the last group's MFMAs, and the first operands of the next step are read right behind it:
		// their LDS latency hides under those MFMAs instead of opening the next step with a stall
		if (!(a.probe & 4u))
			__syncthreads();
		if (k + 1 < steps)
			read_group(buf ^ 1, 0, av0, bv0);
		mfma_group(av1, bv1);
	}

	// epilogue (as k_exact_scores): C[row = (e&3) + 8*(e>>2) + 4*(lane>>5)][col = lane&31]
#pragma unroll
	for (int i = 0; i < 2; ++i) {
#pragma unroll
		for (int j = 0; j < TN; ++j) {
			const uint32_t col = r0 + wn * 32 * TN + j * 32 + (lane & 31);
			const bool col_ok = col < n_rows_total;
			float xn2 = 0.f;
			bool live = false;
			if (col_ok) {
				xn2 = a.row_norm2[col];
				live = a.keys[col] != FREE_KEY;
			}
#pragma unroll
			for (int e = 0; e < 16; ++e) {
				const uint32_t qi = q0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
				if (qi < a.n_queries && col - a.row_begin < a.chunk_stride) {
					const float dot = acc[i][j][e];
					float s;
					if (a.metric == 0)
						s = xn2 - 2.f * dot;
					else if (a.metric == 2)
						s = -dot;
					else
						s = xn2 > 0.f ? -dot * rsqrtf(xn2) : 0.f;
					if (!col_ok || !live)
						s = __builtin_inff();
					if (a.cand_cnt) { // survivors only: what k_exact_select's own threshold test would keep
						const uint32_t ql = qi - q0;
						if (s < 3.0e38f && lex_less(s, col, tau_s[ql], tau_i[ql])) {
							const uint32_t p = atomicAdd(&a.cand_cnt[qi], 1u);
							if (p < a.cand_cap) {
								a.cand_s[(size_t)qi * a.cand_cap + p] = s;
								a.cand_i[(size_t)qi * a.cand_cap + p] = col;
							}
						}
					} else if (!(a.probe & 1u) || s == 12345.678f)
						a.scores[(size_t)qi * a.chunk_stride + (col - a.row_begin)] = s;
				}
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// Round 5: the same 128 x 128 score tile as a PERSISTENT workgroup.  tools/microbench/mfma_f32_peak.hip shows what the chip
// sustains with exactly this inner loop — 64 MFMAs, 16 ds_read_b128 and one barrier per step, two workgroups per compute unit:
// 0.98 of the 157.3 TFLOP/s peak (profiles/r05c_mfma_f32_peak_microbench.txt) — so the 0.76 of k_exact_scores_v2 is the
// kernel's, not the pipe's.  What v2 has and the microbenchmark has not: a prologue (two dependent global round trips before the
// first MFMA) and an epilogue (64 scores per lane, stores or atomics, workgroup exit, dispatch of the successor) every 24
// steps, taken by the two workgroups of a compute unit AT THE SAME TIME — they start together and run identical work, so when
// one cannot feed the matrix pipe neither can the other.  Here:
//   * gridDim.x = two workgroups per compute unit, each walks over tiles w, w + G, w + 2G, ... of the launch; the step
//     sequence is FLATTENED across tiles — the global loads of the next tile's steps 0 and 1 are issued during the last two
//     steps of the current one, its first operands are read before the current tile's last MFMAs — so a tile boundary costs
//     the epilogue's own instructions and nothing else;
//   * (tried: the second workgroup of every compute unit starting half a tile late, so that one workgroup's epilogue falls
//     into the middle of the other's tile — no gain, the pair drifts apart by itself; kept behind probe bit 8);
//   * tile order (x_hi, y, x_lo) with x_lo = 8 consecutive row tiles: with workgroups dealt round-robin over the 8 XCDs, XCD j
//     works on row tiles 8a + j for ALL query tiles y at the same time — the 384 KiB of a row tile are fetched into that
//     XCD's L2 once instead of once per query tile (v2's grid order sent the same row tile to the same XCD 2048 workgroups
//     apart: every query tile re-read the table from HBM);
//   * the barrier is LDS-only (the epilogue's stores / atomics are not waited for at the next step's barrier).
// Same MFMA order per accumulator, same epilogue arithmetic: scores bit-identical to v2's.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_exact_scores_v3(ExactArgs a) {
	using S = X2Shape<2>;
	constexpr int TN = 2, PB = 4;
	extern __shared__ __attribute__((aligned(16))) unsigned char x2_smem[];
	float *const lds = reinterpret_cast<float *>(x2_smem); // [buf][A | B][row][36]
	__shared__ float tau_s[128];
	__shared__ uint32_t tau_i[128];
	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int wm = wave >> 1, wn = wave & 1;
	if (a.cand_cnt && __hip_atomic_load(&a.cand_cnt[a.n_queries], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
		return; // (a filtered pass that has overflowed is repeated the plain way by the host; uniform: only k_exact_select, stream-ordered against this launch, writes the word)
	const uint32_t steps = (a.V + 7) / 8;
	const uint32_t tx = (a.row_end - a.row_begin + 127u) / 128u, ty = (a.n_queries + 127u) / 128u;
	const uint32_t tx8 = tx & ~7u, T = tx * ty, G = gridDim.x, b = blockIdx.x;
	if (b >= T)
		return;
	// item -> (row tile x, query tile y): groups of 8 row tiles under every query tile, the ragged rest in plain order
	auto decode = [&](uint32_t t, uint32_t &x, uint32_t &y) {
		if (t < tx8 * ty) {
			const uint32_t r = t >> 3;
			y = r % ty;
			x = (r / ty) * 8u + (t & 7u);
		} else {
			const uint32_t u = t - tx8 * ty, rem = tx - tx8;
			y = u / rem;
			x = tx8 + u % rem;
		}
	};
	const uint32_t N = ((T - b + G - 1u) / G) * steps; // steps of this workgroup, all its tiles in a row

	f32x16 acc[2][TN];
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int j = 0; j < TN; ++j)
#pragma unroll
			for (int e = 0; e < 16; ++e)
				acc[i][j][e] = 0.f;

	// staging (as v2): thread -> (float4 f of the step's eight, rows rr + 32 p); rows beyond the edge re-read the last one
	const int f = tid & 7, rr = tid >> 3;
	const float4 *qsrc[4], *xsrc[PB];
	uint32_t item_l = b, step_l = 0; // the load cursor: two steps ahead of the MFMAs, possibly in the next tile already
	auto set_load_item = [&](uint32_t t) {
		uint32_t x, y;
		decode(t, x, y);
		const uint32_t q0 = y * 128u, r0 = a.row_begin + x * 128u;
#pragma unroll
		for (int p = 0; p < 4; ++p) {
			uint32_t qi = q0 + rr + 32 * p;
			qi = qi < a.n_queries ? qi : a.n_queries - 1;
			qsrc[p] = a.queries + (size_t)qi * a.V;
		}
#pragma unroll
		for (int p = 0; p < PB; ++p) {
			uint32_t ri = r0 + rr + 32 * p;
			ri = ri < a.row_end ? ri : a.row_end - 1;
			xsrc[p] = a.vectors + (size_t)ri * a.V;
		}
	};
	float4 qa[4], xb[PB];
	auto load_next = [&]() { // the step under the load cursor -> registers (unconditional loads, clamped chunk, zeroed afterwards)
		const uint32_t c = step_l * 8 + f;
		const uint32_t cc = c < a.V ? c : a.V - 1;
#pragma unroll
		for (int p = 0; p < 4; ++p)
			qa[p] = qsrc[p][cc];
#pragma unroll
		for (int p = 0; p < PB; ++p)
			xb[p] = xsrc[p][cc];
		if (c >= a.V) {
#pragma unroll
			for (int p = 0; p < 4; ++p)
				qa[p] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
			for (int p = 0; p < PB; ++p)
				xb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
		}
		if (++step_l == steps) {
			step_l = 0;
			item_l += G;
			if (item_l < T)
				set_load_item(item_l);
		}
	};
	auto store_step = [&](int buf) {
		float *A = lds + buf * (S::A_TILE + S::B_TILE), *B = A + S::A_TILE;
#pragma unroll
		for (int p = 0; p < 4; ++p)
			*reinterpret_cast<float4 *>(A + (rr + 32 * p) * X2_LD + 4 * f) = qa[p];
#pragma unroll
		for (int p = 0; p < PB; ++p)
			*reinterpret_cast<float4 *>(B + (rr + 32 * p) * X2_LD + 4 * f) = xb[p];
	};
	const int a_off = (wm * 64 + (lane & 31)) * X2_LD + 4 * (lane >> 5);
	const int b_off = (wn * 32 * TN + (lane & 31)) * X2_LD + 4 * (lane >> 5);
	auto read_group = [&](int buf, int g, float4 (&av)[2], float4 (&bv)[TN]) {
		const float *A = lds + buf * (S::A_TILE + S::B_TILE), *B = A + S::A_TILE;
#pragma unroll
		for (int i = 0; i < 2; ++i)
			av[i] = *reinterpret_cast<const float4 *>(A + a_off + i * 32 * X2_LD + g * 8);
#pragma unroll
		for (int j = 0; j < TN; ++j)
			bv[j] = *reinterpret_cast<const float4 *>(B + b_off + j * 32 * X2_LD + g * 8);
	};
	auto mfma_group = [&](const float4 (&av)[2], const float4 (&bv)[TN]) {
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
				acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
				acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
				acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
				acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
	};

	// probe bit 8: the compute unit's second workgroup starts half a tile late, so that one workgroup's epilogue falls into the
	// middle of the other's tile.  Measured: no gain (profiles/r05h_*: 131.5 against 130.9 TFLOP/s WITHOUT it) — the two
	// workgroups drift apart on their own — so it is off by default
	if (b >= (G + 1u) / 2u && (a.probe & 8u)) {
		// half a tile = steps / 2 steps of 64 MFMAs x 64 cycles each, shared with the other workgroup: ~ steps * 4096 cycles
		const unsigned long long until = __builtin_readcyclecounter() + (unsigned long long)steps * 4096ull;
		while (__builtin_readcyclecounter() < until)
			__builtin_amdgcn_s_sleep(64);
	}

	// prologue: this workgroup's first step into buffer 0, the second on its way
	set_load_item(b);
	load_next();
	store_step(0);
	if (N > 1)
		load_next();
	__syncthreads();
	float4 av0[2], bv0[TN], av1[2], bv1[TN];
	read_group(0, 0, av0, bv0);
	uint32_t item_c = b, step_c = 0; // the compute cursor
	// (declared outside the loop and written only in a tile's last step: re-initialising them every step made hipcc wait for
	//  ALL global loads in flight — the prefetch of two steps ahead — at the top of every step, a write-after-write guard)
	uint32_t cx = 0, cy = 0;
	float my_tau_s = 0.f;
	uint32_t my_tau_i = 0;
	float col_n2[TN];
	int64_t col_key[TN];
#pragma unroll
	for (int j = 0; j < TN; ++j)
		col_n2[j] = 0.f, col_key[j] = FREE_KEY;
	for (uint32_t g = 0; g < N; ++g) {
		const int buf = (int)(g & 1u);
		const bool last = step_c + 1 == steps;
		if (last) {
			// (wave-uniform) this tile's coordinates; what its epilogue reads from memory — the norms and keys of this lane's two
			// columns and, filtered, the thresholds of the tile's queries — is asked for now, a whole step ahead (v2's epilogue
			// paid four dependent round trips for them)
			decode(item_c, cx, cy);
#pragma unroll
			for (int j = 0; j < TN; ++j) {
				uint32_t col = a.row_begin + cx * 128u + wn * 32 * TN + j * 32 + (lane & 31);
				col = col < a.row_end ? col : a.row_end - 1;
				col_n2[j] = a.row_norm2[col];
				col_key[j] = a.keys[col];
			}
			if (a.cand_cnt && tid < 128) {
				const uint32_t qi = cy * 128u + tid < a.n_queries ? cy * 128u + tid : a.n_queries - 1;
				my_tau_s = a.best_s[(size_t)qi * a.KP + a.KP - 1];
				my_tau_i = a.best_i[(size_t)qi * a.KP + a.KP - 1];
			}
		}
		read_group(buf, 1, av1, bv1);
		mfma_group(av0, bv0);
		read_group(buf, 2, av0, bv0);
		if (g + 1 < N && !(a.probe & 16u))
			store_step(buf ^ 1); // the registers of the next step (loaded a whole step ago) go to the idle buffer
		mfma_group(av1, bv1);
		read_group(buf, 3, av1, bv1);
		if (g + 2 < N && !(a.probe & 2u))
			load_next(); // lands during the next step's MFMAs
		mfma_group(av0, bv0);
		if (!(a.probe & 4u))
			lds_barrier();
		if (g + 1 < N)
			read_group(buf ^ 1, 0, av0, bv0);
		mfma_group(av1, bv1);
		if (!last || (a.probe & 32u)) {
			if (last)
				item_c += G, step_c = 0;
			else
				++step_c;
			continue;
		}
		// ---- epilogue of tile (cx, cy): C[row = (e&3) + 8*(e>>2) + 4*(lane>>5)][col = lane&31]
		const uint32_t q0 = cy * 128u, r0 = a.row_begin + cx * 128u;
		if (a.cand_cnt) {
			if (tid < 128)
				tau_s[tid] = my_tau_s, tau_i[tid] = my_tau_i;
			lds_barrier();
		}
#pragma unroll
		for (int i = 0; i < 2; ++i) {
#pragma unroll
			for (int j = 0; j < TN; ++j) {
				const uint32_t col = r0 + wn * 32 * TN + j * 32 + (lane & 31);
				const bool col_ok = col < a.row_end;
				const float xn2 = col_n2[j];
				const bool live = col_key[j] != FREE_KEY;
#pragma unroll
				for (int e = 0; e < 16; ++e) {
					const uint32_t qi = q0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
					if (qi < a.n_queries && col - a.row_begin < a.chunk_stride) {
						const float dot = acc[i][j][e];
						float s;
						if (a.metric == 0)
							s = xn2 - 2.f * dot;
						else if (a.metric == 2)
							s = -dot;
						else
							s = xn2 > 0.f ? -dot * rsqrtf(xn2) : 0.f;
						if (!col_ok || !live)
							s = __builtin_inff();
						if (a.cand_cnt) { // survivors only: what k_exact_select's own threshold test would keep
							const uint32_t ql = qi - q0;
							if (s < 3.0e38f && lex_less(s, col, tau_s[ql], tau_i[ql])) {
								const uint32_t p = atomicAdd(&a.cand_cnt[qi], 1u);
								if (p < a.cand_cap) {
									a.cand_s[(size_t)qi * a.cand_cap + p] = s;
									a.cand_i[(size_t)qi * a.cand_cap + p] = col;
								}
							}
						} else if (!(a.probe & 1u) || s == 12345.678f)
							a.scores[(size_t)qi * a.chunk_stride + (col - a.row_begin)] = s;
					}
					acc[i][j][e] = 0.f;
				}
			}
		}
		item_c += G;
		step_c = 0;
	}
}

// ---------------------------------------------------------------------------------------------------------
// Round 5, second step: the persistent tile with its operands brought in by LDS-DMA.  The ablation of k_exact_scores_v3
// (profiles/r05g_exact_tile_ablation_plain_mode.txt: 117 TFLOP/s whole; without the epilogue 128, without global loads 126,
// without either 139, also without the LDS writes 146, also without the barrier 147 — against the 155 the same inner loop
// sustains in the microbenchmark) says where the matrix pipe's idle quarter goes: the staging instructions (8 global loads
// with 64-bit addresses, 8 waits and 8 ds_write_b128 per step and thread, 32 staging registers) and an epilogue of ~60
// instructions per score.  Here:
//   * global_load_lds_dwordx4 (gfx950): a wave's 64 lanes x 16 bytes land in 1 KiB of contiguous LDS, no registers, no
//     ds_write.  The LDS image of a step is therefore lane-linear — [row][8 chunks of 16 B], 128 B per row, no padding — and
//     bank conflicts are avoided by an XOR swizzle applied on BOTH sides (guide rule 21): the lane that fills position p of
//     row r fetches logical chunk p ^ (r & 7) from memory (still the row's full 128-byte line per 8 lanes), the MFMA operand
//     read of logical chunk c of row r goes to position c ^ (r & 7);
//   * addresses: one 64-bit SCALAR base per operand and tile, advanced by 128 bytes per step with scalar adds, plus a 32-bit
//     per-lane offset that is constant for the whole tile (saddr form) — no vector address arithmetic in the loop;
//   * two LDS buffers of 32 KiB: the DMA of step k+1 is issued right after the barrier that ends step k-1 and is waited for
//     (s_waitcnt vmcnt(0), by hand: hipcc does not count asm memory operations) just before the barrier that ends step k;
//   * the epilogue reads its thresholds four at a time (ds_read_b128), tests four scores per branch, keeps all address
//     arithmetic of the common case in registers that are set up once per tile; a ragged tile (fewer than 128 queries, a
//     window that ends inside the tile) takes v3's generic epilogue.
// Needs V % 8 == 0 (whole 128-byte steps: 768, 1536, 128, 1024 ... dimensions; the DMA cannot zero a partial step) — other
// dimensions run k_exact_scores_v3.  Same MFMA order per accumulator, same score arithmetic: bit-identical scores.
constexpr uint32_t X4_IMAGE_BYTES = 128 * 128;                  // one operand of one step: 128 rows x 128 bytes
constexpr uint32_t X4_LDS_BYTES = 2 * 2 * X4_IMAGE_BYTES;        // [buffer][A | B]

__device__ __forceinline__ void glds16(uint32_t lds_dst, uint32_t lane_offset, const void *base) {
	// (M0 = the wave-uniform LDS byte address; the 64 lanes' 16 bytes land at lds_dst + 16 * lane; written and restored in
	//  the same statement: M0 is compiler-reserved and not preserved around asm)
	unsigned keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep)
	             : "v"(lane_offset), "s"(lds_dst), "s"(base)
	             : "memory");
}

template <int MT>
__device__ __forceinline__ float exact_score(float dot, float xn2, float rs) {
	if (MT == 0)
		return xn2 - 2.f * dot;
	if (MT == 2)
		return -dot;
	return xn2 > 0.f ? -dot * rs : 0.f;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_exact_scores_v4(ExactArgs a) {
	constexpr int TN = 2;
	extern __shared__ __attribute__((aligned(16))) unsigned char x2_smem[];
	__shared__ __attribute__((aligned(16))) float tau_s[128];
	__shared__ __attribute__((aligned(16))) uint32_t tau_i[128];
	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1;
	if (a.cand_cnt && __hip_atomic_load(&a.cand_cnt[a.n_queries], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
		return; // (a filtered pass that has overflowed is repeated the plain way by the host; uniform: only k_exact_select, stream-ordered against this launch, writes the word)
	const uint32_t steps = a.V / 8; // (V % 8 == 0: the host's choice of this kernel)
	const uint32_t tx = (a.row_end - a.row_begin + 127u) / 128u, ty = (a.n_queries + 127u) / 128u;
	const uint32_t tx8 = tx & ~7u, T = tx * ty, G = gridDim.x, b = blockIdx.x;
	if (b >= T)
		return;
	auto decode = [&](uint32_t t, uint32_t &x, uint32_t &y) { // as k_exact_scores_v3
		if (t < tx8 * ty) {
			const uint32_t r = t >> 3;
			y = r % ty;
			x = (r / ty) * 8u + (t & 7u);
		} else {
			const uint32_t u = t - tx8 * ty, rem = tx - tx8;
			y = u / rem;
			x = tx8 + u % rem;
		}
	};
	const uint32_t N = ((T - b + G - 1u) / G) * steps;
	const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)x2_smem);

	f32x16 acc[2][TN];
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int j = 0; j < TN; ++j)
#pragma unroll
			for (int e = 0; e < 16; ++e)
				acc[i][j][e] = 0.f;

	// ---- the load side.  DMA instruction p (0..3) of wave w fills rows 8 (4 w + p) .. + 7 of an operand image: lane l is
	// position l & 7 of row 8 (4 w + p) + (l >> 3) and fetches logical chunk (l & 7) ^ (l >> 3) of that row (row & 7 = l >> 3)
	const uint32_t row_in_piece = (uint32_t)lane >> 3, chunk = ((uint32_t)lane & 7u) ^ row_in_piece;
	uint32_t off_a[4], off_b[4]; // byte offsets from the tile's scalar bases; constant over the tile's steps
	const unsigned char *base_a = nullptr, *base_b = nullptr; // wave-uniform: tile origin + 128 bytes per step
	uint32_t item_l = b, step_l = 0;
	auto set_load_item = [&](uint32_t t) {
		uint32_t x, y;
		decode(t, x, y);
		const uint32_t q0 = y * 128u, r0 = a.row_begin + x * 128u;
		base_a = reinterpret_cast<const unsigned char *>(a.queries + (size_t)q0 * a.V);
		base_b = reinterpret_cast<const unsigned char *>(a.vectors + (size_t)r0 * a.V);
#pragma unroll
		for (int p = 0; p < 4; ++p) {
			const uint32_t r = 8u * (4u * (uint32_t)wave + p) + row_in_piece;
			const uint32_t qr = q0 + r < a.n_queries ? r : a.n_queries - 1 - q0; // rows beyond the edge re-read the last one
			const uint32_t xr = r0 + r < a.row_end ? r : a.row_end - 1 - r0;
			off_a[p] = (qr * a.V + chunk) * 16u;
			off_b[p] = (xr * a.V + chunk) * 16u;
		}
	};
	auto dma_next = [&](int buf) { // the step under the load cursor -> LDS buffer `buf`, asynchronously; advances the cursor
		const uint32_t dst = lds0 + (uint32_t)buf * 2u * X4_IMAGE_BYTES + (uint32_t)wave * 4096u;
#pragma unroll
		for (int p = 0; p < 4; ++p)
			glds16(dst + p * 1024u, off_a[p], base_a);
#pragma unroll
		for (int p = 0; p < 4; ++p)
			glds16(dst + X4_IMAGE_BYTES + p * 1024u, off_b[p], base_b);
		base_a += 128, base_b += 128;
		if (++step_l == steps) {
			step_l = 0;
			item_l += G;
			if (item_l < T)
				set_load_item(item_l);
		}
	};
	// ---- the MFMA side: row (lane & 31) of each 32-row tile, logical chunk 2 g + (lane >> 5) of k-group g, at position
	// chunk ^ (row & 7) = chunk ^ (lane & 7)
	const unsigned char *const lds = x2_smem;
	const uint32_t a_row = (uint32_t)(wm * 64 + (lane & 31)) * 128u, b_row = X4_IMAGE_BYTES + (uint32_t)(wn * 32 * TN + (lane & 31)) * 128u;
	uint32_t pos[4];
#pragma unroll
	for (int g = 0; g < 4; ++g)
		pos[g] = (((uint32_t)(2 * g) + ((uint32_t)lane >> 5)) ^ ((uint32_t)lane & 7u)) * 16u;
	auto read_group = [&](int buf, int g, float4 (&av)[2], float4 (&bv)[TN]) {
		const unsigned char *B = lds + buf * 2 * X4_IMAGE_BYTES;
#pragma unroll
		for (int i = 0; i < 2; ++i)
			av[i] = *reinterpret_cast<const float4 *>(B + a_row + i * 32 * 128 + pos[g]);
#pragma unroll
		for (int j = 0; j < TN; ++j)
			bv[j] = *reinterpret_cast<const float4 *>(B + b_row + j * 32 * 128 + pos[g]);
	};
	auto mfma_group = [&](const float4 (&av)[2], const float4 (&bv)[TN]) {
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
				acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
				acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
				acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
				acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
	};
	// this wave's DMA has landed and its own LDS reads have returned; then everybody's
	auto step_barrier = [&] { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

	if (b >= (G + 1u) / 2u && (a.probe & 8u)) { // probe: the compute unit's second workgroup half a tile late (as v3: no gain)
		const unsigned long long until = __builtin_readcyclecounter() + (unsigned long long)steps * 4096ull;
		while (__builtin_readcyclecounter() < until)
			__builtin_amdgcn_s_sleep(64);
	}

	set_load_item(b);
	dma_next(0);
	step_barrier();
	float4 av0[2], bv0[TN], av1[2], bv1[TN];
	read_group(0, 0, av0, bv0);
	uint32_t item_c = b, step_c = 0;
	uint32_t cx = 0, cy = 0; // (written only in a tile's last step, as everything its epilogue reads from memory: see v3)
	float my_tau_s = 0.f;
	uint32_t my_tau_i = 0;
	float col_n2[TN];
	int64_t col_key[TN];
#pragma unroll
	for (int j = 0; j < TN; ++j)
		col_n2[j] = 0.f, col_key[j] = FREE_KEY;
	for (uint32_t g = 0; g < N; ++g) {
		const int buf = (int)(g & 1u);
		const bool last = step_c + 1 == steps;
		if (g + 1 < N && !(a.probe & 2u))
			dma_next(buf ^ 1); // (everybody has finished reading that buffer: the barrier that ended the previous step)
		if (last) {
			decode(item_c, cx, cy);
#pragma unroll
			for (int j = 0; j < TN; ++j) {
				uint32_t col = a.row_begin + cx * 128u + wn * 32 * TN + j * 32 + (lane & 31);
				col = col < a.row_end ? col : a.row_end - 1;
				col_n2[j] = a.row_norm2[col];
				col_key[j] = a.keys[col];
			}
			if (a.cand_cnt && tid < 128) {
				const uint32_t qi = cy * 128u + tid < a.n_queries ? cy * 128u + tid : a.n_queries - 1;
				my_tau_s = a.best_s[(size_t)qi * a.KP + a.KP - 1];
				my_tau_i = a.best_i[(size_t)qi * a.KP + a.KP - 1];
			}
		}
		read_group(buf, 1, av1, bv1);
		mfma_group(av0, bv0);
		read_group(buf, 2, av0, bv0);
		mfma_group(av1, bv1);
		read_group(buf, 3, av1, bv1);
		mfma_group(av0, bv0);
		if (!(a.probe & 4u))
			step_barrier();
		if (g + 1 < N)
			read_group(buf ^ 1, 0, av0, bv0);
		mfma_group(av1, bv1);
		if (!last || (a.probe & 32u)) {
			if (last)
				item_c += G, step_c = 0;
			else
				++step_c;
			continue;
		}
		// ---- epilogue of tile (cx, cy): C[row = (e&3) + 8*(e>>2) + 4*(lane>>5)][col = lane&31]
		const uint32_t q0 = cy * 128u, r0 = a.row_begin + cx * 128u;
		if (a.cand_cnt) {
			if (tid < 128)
				tau_s[tid] = my_tau_s, tau_i[tid] = my_tau_i;
			lds_barrier();
		}
		const bool whole = q0 + 128u <= a.n_queries && r0 + 128u <= a.row_end && r0 + 128u - a.row_begin <= a.chunk_stride;
		if (whole) {
			// the common case: every query of the tile exists, every column lies inside the window
			const uint32_t qrow = q0 + wm * 64 + 4 * ((uint32_t)lane >> 5); // + 32 i + (e & 3) + 8 (e >> 2)
			auto run = [&](auto mt) {
				constexpr int MT = decltype(mt)::value;
#pragma unroll
				for (int j = 0; j < TN; ++j) {
					const uint32_t col = r0 + wn * 32 * TN + j * 32 + (lane & 31);
					const float xn2 = col_n2[j], rs = MT == 1 ? rsqrtf(col_n2[j]) : 0.f;
					const bool dead = col_key[j] == FREE_KEY;
					if (a.cand_cnt) {
#pragma unroll
						for (int i = 0; i < 2; ++i) {
#pragma unroll
							for (int e4 = 0; e4 < 4; ++e4) {
								const uint32_t ql = wm * 64 + i * 32 + 8 * e4 + 4 * ((uint32_t)lane >> 5);
								const float4 ts = *reinterpret_cast<const float4 *>(&tau_s[ql]);
								const uint4 ti = *reinterpret_cast<const uint4 *>(&tau_i[ql]);
								float sc[4];
#pragma unroll
								for (int u = 0; u < 4; ++u)
									sc[u] = dead ? __builtin_inff() : exact_score<MT>(acc[i][j][4 * e4 + u], xn2, rs);
								const bool k0 = sc[0] < 3.0e38f && lex_less(sc[0], col, ts.x, ti.x), k1 = sc[1] < 3.0e38f && lex_less(sc[1], col, ts.y, ti.y);
								const bool k2 = sc[2] < 3.0e38f && lex_less(sc[2], col, ts.z, ti.z), k3 = sc[3] < 3.0e38f && lex_less(sc[3], col, ts.w, ti.w);
								if (k0 | k1 | k2 | k3) { // survivors are rare: a few dozen per query and launch
									const bool keep[4] = {k0, k1, k2, k3};
#pragma unroll
									for (int u = 0; u < 4; ++u) {
										if (keep[u]) {
											const uint32_t qi = q0 + ql + u;
											const uint32_t p = atomicAdd(&a.cand_cnt[qi], 1u);
											if (p < a.cand_cap) {
												a.cand_s[(size_t)qi * a.cand_cap + p] = sc[u];
												a.cand_i[(size_t)qi * a.cand_cap + p] = col;
											}
										}
									}
								}
							}
						}
					} else if (!(a.probe & 1u)) {
						float *out = a.scores + (size_t)qrow * a.chunk_stride + (col - a.row_begin);
#pragma unroll
						for (int i = 0; i < 2; ++i)
#pragma unroll
							for (int e = 0; e < 16; ++e)
								out[(size_t)(i * 32 + (e & 3) + 8 * (e >> 2)) * a.chunk_stride] =
								    dead ? __builtin_inff() : exact_score<MT>(acc[i][j][e], xn2, rs);
					}
				}
			};
			if (a.metric == 0)
				run(std::integral_constant<int, 0> {});
			else if (a.metric == 2)
				run(std::integral_constant<int, 2> {});
			else
				run(std::integral_constant<int, 1> {});
		} else {
#pragma unroll
			for (int i = 0; i < 2; ++i) {
#pragma unroll
				for (int j = 0; j < TN; ++j) {
					const uint32_t col = r0 + wn * 32 * TN + j * 32 + (lane & 31);
					const bool col_ok = col < a.row_end;
					const float xn2 = col_n2[j];
					const bool live = col_key[j] != FREE_KEY;
#pragma unroll
					for (int e = 0; e < 16; ++e) {
						const uint32_t qi = q0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
						if (qi < a.n_queries && col - a.row_begin < a.chunk_stride) {
							const float dot = acc[i][j][e];
							float s;
							if (a.metric == 0)
								s = xn2 - 2.f * dot;
							else if (a.metric == 2)
								s = -dot;
							else
								s = xn2 > 0.f ? -dot * rsqrtf(xn2) : 0.f;
							if (!col_ok || !live)
								s = __builtin_inff();
							if (a.cand_cnt) {
								const uint32_t ql = qi - q0;
								if (s < 3.0e38f && lex_less(s, col, tau_s[ql], tau_i[ql])) {
									const uint32_t p = atomicAdd(&a.cand_cnt[qi], 1u);
									if (p < a.cand_cap) {
										a.cand_s[(size_t)qi * a.cand_cap + p] = s;
										a.cand_i[(size_t)qi * a.cand_cap + p] = col;
									}
								}
							} else if (!(a.probe & 1u) || s == 12345.678f)
								a.scores[(size_t)qi * a.chunk_stride + (col - a.row_begin)] = s;
						}
					}
				}
			}
		}
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
#pragma unroll
				for (int e = 0; e < 16; ++e)
					acc[i][j][e] = 0.f;
		item_c += G;
		step_c = 0;
	}
}

// ---------------------------------------------------------------------------------------------------------
// Running top-K' per query.  best_s / best_i: n_queries x KP, ascending by (score, index); unused = (+inf, EMPTY).
struct SelectArgs {
	const float *scores;
	uint32_t chunk_stride;
	uint32_t chunk_cols; // valid columns in this chunk
	uint32_t row_begin;
	uint32_t KP;
	float *best_s;
	uint32_t *best_i;
	// filtered mode (scores == NULL): the survivors the score tiles appended since the last select
	uint32_t cand_cap;
	uint32_t *cand_cnt;
	const float *cand_s;
	const uint32_t *cand_i;
	uint32_t *overflow; // set when a query collected more survivors than its buffer holds (the host reruns the plain way)
};

constexpr int SEL_THREADS = 256;
constexpr int SEL_CAP = 2048;
constexpr int SEL_KP_MAX = 4096; // largest k + slack of the running top-K' (32 KiB of LDS; covers every LIMIT the reference's
                                 // top-k rewrite accepts, k < 2048: hnsw_optimize_topk.cpp:170-173)

__device__ __forceinline__ void block_argmin(float &s, uint32_t &i, float *red_s, uint32_t *red_i) {
	// wave reduce
	for (int o = 32; o >= 1; o >>= 1) {
		float os = __shfl_xor(s, o);
		uint32_t oi = __shfl_xor(i, o);
		if (lex_less(os, oi, s, i))
			s = os, i = oi;
	}
	const int w = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0)
		red_s[w] = s, red_i[w] = i;
	__syncthreads();
	s = red_s[0], i = red_i[0];
	for (int k = 1; k < SEL_THREADS / 64; ++k)
		if (lex_less(red_s[k], red_i[k], s, i))
			s = red_s[k], i = red_i[k];
	__syncthreads();
}

__global__ __launch_bounds__(SEL_THREADS) void k_exact_select(SelectArgs a) {
	__shared__ float buf_s[SEL_CAP];
	__shared__ uint32_t buf_i[SEL_CAP];
	__shared__ float old_s[SEL_KP_MAX];
	__shared__ uint32_t old_i[SEL_KP_MAX];
	__shared__ float red_s[SEL_THREADS / 64];
	__shared__ uint32_t red_i[SEL_THREADS / 64];
	__shared__ uint32_t cnt;
	const uint32_t q = blockIdx.x;
	const int tid = threadIdx.x;
	// An earlier select of this filtered pass gave up: the host discards everything, this launch has nothing to add.  Another
	// workgroup of THIS launch may raise the flag at any moment, so the decision is taken once per workgroup — one thread reads
	// the word, everybody branches on the copy behind a barrier — and never by some waves only ahead of the barriers below
	// (ADVICE r05).
	__shared__ uint32_t gave_up;
	if (tid == 0)
		gave_up = a.scores ? 0u : __hip_atomic_load(a.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__syncthreads();
	if (gave_up)
		return;
	const float *row = a.scores + (size_t)q * a.chunk_stride;
	float *bs = a.best_s + (size_t)q * a.KP;
	uint32_t *bi = a.best_i + (size_t)q * a.KP;
	for (uint32_t i = tid; i < a.KP; i += SEL_THREADS)
		old_s[i] = bs[i], old_i[i] = bi[i];
	if (tid == 0)
		cnt = 0;
	__syncthreads();
	const float tau_s = old_s[a.KP - 1];
	const uint32_t tau_i = old_i[a.KP - 1];
	if (!a.scores) { // filtered mode: the score tiles have already applied the threshold (an older, hence looser, one)
		const uint32_t have = a.cand_cnt[q];
		if (have > a.cand_cap || have > (uint32_t)SEL_CAP) {
			if (tid == 0)
				*a.overflow = 1u;
			return; // (the host discards everything and reruns with the scores stored)
		}
		for (uint32_t c = tid; c < have; c += SEL_THREADS) {
			const float s = a.cand_s[(size_t)q * a.cand_cap + c];
			const uint32_t idx = a.cand_i[(size_t)q * a.cand_cap + c];
			if (lex_less(s, idx, tau_s, tau_i)) {
				const uint32_t p = atomicAdd(&cnt, 1u);
				buf_s[p] = s, buf_i[p] = idx;
			}
		}
		__syncthreads();
		if (tid == 0)
			a.cand_cnt[q] = 0; // consumed
	} else {
		for (uint32_t c = tid; c < a.chunk_cols; c += SEL_THREADS) {
			const float s = row[c];
			const uint32_t idx = a.row_begin + c;
			if (s < 3.0e38f && lex_less(s, idx, tau_s, tau_i)) {
				const uint32_t p = atomicAdd(&cnt, 1u);
				if (p < SEL_CAP)
					buf_s[p] = s, buf_i[p] = idx;
			}
		}
		__syncthreads();
	}
	const uint32_t n = cnt;
	if (n == 0)
		return;
	const bool overflow = n > SEL_CAP;
	// successive minima over (old top-K') U (buffer | whole chunk), strictly above the last one taken
	float last_s = -__builtin_inff();
	uint32_t last_i = 0;
	bool first = true;
	for (uint32_t out = 0; out < a.KP; ++out) {
		float s = __builtin_inff();
		uint32_t i = EMPTY_SLOT;
		auto consider = [&](float cs, uint32_t ci) {
			if ((first || lex_less(last_s, last_i, cs, ci)) && lex_less(cs, ci, s, i))
				s = cs, i = ci;
		};
		for (uint32_t j = tid; j < a.KP; j += SEL_THREADS)
			consider(old_s[j], old_i[j]);
		if (!overflow) {
			for (uint32_t j = tid; j < n; j += SEL_THREADS)
				consider(buf_s[j], buf_i[j]);
		} else {
			for (uint32_t c = tid; c < a.chunk_cols; c += SEL_THREADS)
				consider(row[c], a.row_begin + c);
		}
		block_argmin(s, i, red_s, red_i);
		if (tid == 0)
			bs[out] = s, bi[out] = i;
		if (i == EMPTY_SLOT) { // nothing left: the rest stays (+inf, EMPTY)
			for (uint32_t j = out + 1 + tid; j < a.KP; j += SEL_THREADS)
				bs[j] = __builtin_inff(), bi[j] = EMPTY_SLOT;
			break;
		}
		last_s = s, last_i = i, first = false;
	}
}

#endif // VSS_ENGINE_TU

// ---------------------------------------------------------------------------------------------------------
// Exact re-rank of the K' survivors: one wave per query.
struct RerankArgs {
	GraphView gv;
	const float *queries; // n_queries x q_stride
	uint32_t q_stride;
	uint32_t n_queries;
	uint32_t k, KP;
	const uint32_t *best_i;
	int64_t *out_keys;
	float *out_d;
	uint32_t *out_count;
};

template <int MT, int NCH, int R>
__global__ __launch_bounds__(64) void k_exact_rerank(RerankArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = lane_id();
	const uint32_t qi = blockIdx.x;
	float4 *q = reinterpret_cast<float4 *>(smem);
	uint32_t *ids = reinterpret_cast<uint32_t *>(smem + align16(a.gv.sp.V * 16));
	float *dist = reinterpret_cast<float *>(smem + align16(a.gv.sp.V * 16) + align16(a.KP * 4));
	stage_query(q, a.queries + (size_t)qi * a.q_stride, a.gv.dim, a.gv.sp.V);
	const float qa2 = MT == 1 ? wave_query_norm(a.gv.sp, q) : 0.f;
	// compact the valid candidates
	int n = 0;
	for (uint32_t off = 0; off < a.KP; off += 64) {
		uint32_t id = (off + lane < a.KP) ? a.best_i[(size_t)qi * a.KP + off + lane] : EMPTY_SLOT;
		unsigned long long m = __ballot(id != EMPTY_SLOT);
		if (id != EMPTY_SLOT)
			ids[n + __popcll(m & lanes_below(lane))] = id;
		n += __popcll(m);
	}
	wave_sync();
	wave_distances<MT, NCH, R>(a.gv.sp, q, qa2, ids, n, dist);
	// rank by (distance, slot) and emit the first k
	const int count = n < (int)a.k ? n : (int)a.k;
	for (int i = lane; i < n; i += 64) {
		const float di = dist[i];
		const uint32_t si = ids[i];
		int rank = 0;
		for (int j = 0; j < n; ++j)
			rank += lex_less(dist[j], ids[j], di, si);
		if (rank < count) {
			a.out_keys[(size_t)qi * a.k + rank] = a.gv.keys[si];
			if (a.out_d)
				a.out_d[(size_t)qi * a.k + rank] = di;
		}
	}
	for (int i = count + lane; i < (int)a.k; i += 64) {
		a.out_keys[(size_t)qi * a.k + i] = -1ll;
		if (a.out_d)
			a.out_d[(size_t)qi * a.k + i] = __builtin_inff();
	}
	if (lane == 0)
		a.out_count[qi] = count;
}

#ifdef VSS_ENGINE_TU
// ---------------------------------------------------------------------------------------------------------
// array_distance / array_cosine_distance / array_negative_inner_product (DuckDB core scalar functions named at
// reference hnsw_index.cpp:659-673).  One G-lane group per row; rows are `dim` contiguous floats (no padding —
// the ARRAY child vector), so the float4 path is taken only when dim % 4 == 0 and the bases are 16-byte aligned.
template <bool VEC4>
__global__ void k_array_distance(int fn, const float *A, const float *Bm, int b_const, uint64_t rows, uint32_t dim,
                                 uint32_t G, uint32_t logG, float *out) {
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t g = lane & (G - 1), sub = lane >> logG, RG = 64 >> logG;
	const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
	for (uint64_t base = wave * RG; base < rows; base += n_waves * RG) {
		const uint64_t row = base + sub;
		float ab = 0.f, a2 = 0.f, b2 = 0.f;
		if (row < rows) {
			const float *ap = A + row * dim;
			const float *bp = b_const ? Bm : Bm + row * dim;
			if (VEC4) {
				const uint32_t V = dim >> 2;
				const float4 *a4 = reinterpret_cast<const float4 *>(ap);
				const float4 *b4 = reinterpret_cast<const float4 *>(bp);
				for (uint32_t c = g; c < V; c += G) {
					const float4 x = a4[c], y = b4[c];
					if (fn == 0) {
						float t;
						t = x.x - y.x, ab = __fmaf_rn(t, t, ab);
						t = x.y - y.y, ab = __fmaf_rn(t, t, ab);
						t = x.z - y.z, ab = __fmaf_rn(t, t, ab);
						t = x.w - y.w, ab = __fmaf_rn(t, t, ab);
					} else {
						ab = __fmaf_rn(x.x, y.x, ab), ab = __fmaf_rn(x.y, y.y, ab);
						ab = __fmaf_rn(x.z, y.z, ab), ab = __fmaf_rn(x.w, y.w, ab);
						if (fn == 1) {
							a2 = __fmaf_rn(x.x, x.x, a2), a2 = __fmaf_rn(x.y, x.y, a2);
							a2 = __fmaf_rn(x.z, x.z, a2), a2 = __fmaf_rn(x.w, x.w, a2);
							b2 = __fmaf_rn(y.x, y.x, b2), b2 = __fmaf_rn(y.y, y.y, b2);
							b2 = __fmaf_rn(y.z, y.z, b2), b2 = __fmaf_rn(y.w, y.w, b2);
						}
					}
				}
			} else {
				for (uint32_t i = g; i < dim; i += G) {
					const float x = ap[i], y = bp[i];
					if (fn == 0) {
						const float t = x - y;
						ab = __fmaf_rn(t, t, ab);
					} else {
						ab = __fmaf_rn(x, y, ab);
						if (fn == 1)
							a2 = __fmaf_rn(x, x, a2), b2 = __fmaf_rn(y, y, b2);
					}
				}
			}
		}
		ab = group_butterfly(ab, G);
		if (fn == 1) {
			a2 = group_butterfly(a2, G);
			b2 = group_butterfly(b2, G);
		}
		if (g == 0 && row < rows) {
			float r;
			if (fn == 0)
				r = vss_sqrt(ab);
			else if (fn == 2)
				r = -ab;
			else {
				// cosine similarity clamped to [-1, 1]; the comparisons are false for NaN, so a zero
				// norm (0 / 0) or a NaN / inf element yields NaN, never a number that looks like a distance
				float sim = __fdiv_rn(ab, vss_sqrt(__fmul_rn(a2, b2)));
				sim = sim > 1.0f ? 1.0f : (sim < -1.0f ? -1.0f : sim);
				r = 1.0f - sim;
			}
			out[row] = r;
		}
	}
}

// (Round 4, measured and not kept: a variant with the chunk count as a template parameter and 4 / 2 rows per wave in flight —
// every load of a pass issued before the first FMA — runs the 4M x 768 column at the same 0.75-0.78 of the HBM peak as this
// kernel in the same session: profiles/r04i_array_functions_rows_in_flight_not_kept.json.  Box-to-box spread of this kernel:
// 0.69-0.78.)

// ---------------------------------------------------------------------------------------------------------
// Merge of per-shard top-k lists: one wave per query.
// Shard `sh` contributes in_d[sh * stride_d + q * k + j] / in_id[sh * stride_id + q * k + j] (strides in elements): the
// plain layout has both strides = n_queries * k; the packed layout of one all-gather per launch (row ids of a rank's
// whole launch, then its distances, in one block per rank) passes the block size instead.
//
// Output order = ascending (distance, row id) over the valid cells (row id >= 0) of the union — dump_to's order over the
// union (reference hnsw_index.cpp:333-339).  Round 6: a k-way merge by co-ranking instead of rank-by-counting over all
// (G k)^2 pairs with two global loads each (640 000 iterations per query at 8 x 100).  The contract hands over lists that are
// ascending per shard with the unused cells (row id -1) at the tail, so the rank of an entry is a sum over the shards of
//     lower_bound(shard, d) + #{cells of that shard's run of EQUAL distances with a smaller row id}:
// log2(k) probes per shard instead of k.  STAGED: the query's G x k cells are read from HBM once, coalesced, into LDS
// (12 bytes per cell: 9.6 kB at 8 x 100) and probed there; shapes beyond the LDS budget (8 x 2047 = 196 kB) probe the
// global arrays directly (L2).  Whether the lists really are ascending with a packed tail is CHECKED while they are read;
// a query whose lists are not takes the counting path below, so the answer does not depend on the promise.
constexpr uint32_t MERGE_STAGE_MAX_BYTES = 48u * 1024u; // per query (= per 64-thread workgroup): 3 resident per 160 KiB at worst

template <bool STAGED>
struct MergeCells {
	const float *d;      // STAGED: LDS [n_shards][k]; else the global array, already offset to the query's first cell
	const int64_t *id;
	size_t stride_d, stride_id; // elements between shards
	__device__ __forceinline__ float dist(uint32_t sh, uint32_t j) const {
		return d[sh * stride_d + j];
	}
	__device__ __forceinline__ int64_t key(uint32_t sh, uint32_t j) const {
		return id[sh * stride_id + j];
	}
};

// entries of shard `sh` (its first `nv` cells are the valid ones, ascending) that come before (di, idi)
template <bool STAGED>
__device__ __forceinline__ uint32_t merge_cells_before(const MergeCells<STAGED> &c, uint32_t sh, uint32_t nv, float di,
                                                       int64_t idi) {
	uint32_t lo = 0, hi = nv; // first cell with distance >= di
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (c.dist(sh, mid) < di)
			lo = mid + 1;
		else
			hi = mid;
	}
	uint32_t n = lo;
	for (uint32_t j = lo; j < nv && c.dist(sh, j) == di; ++j) // the run of equal distances: the row id decides
		n += c.key(sh, j) < idi ? 1u : 0u;
	return n;
}

template <bool STAGED>
__global__ __launch_bounds__(64) void k_merge_topk(const float *in_d, const int64_t *in_id, size_t stride_d, size_t stride_id,
                                                   uint32_t n_shards, uint32_t n_queries, uint32_t k, float *out_d,
                                                   int64_t *out_id, uint32_t *out_count) {
	extern __shared__ __attribute__((aligned(16))) unsigned char merge_smem[];
	const int lane = threadIdx.x & 63;
	const uint32_t q = blockIdx.x;
	const uint32_t total = n_shards * k;
	const size_t first = (size_t)q * k;
	// LDS: [n_shards] valid counts, then (STAGED) the row ids and the distances of the query's cells
	uint32_t *nvalid = reinterpret_cast<uint32_t *>(merge_smem);
	int64_t *s_id = reinterpret_cast<int64_t *>(merge_smem + ((n_shards * 4u + 15u) & ~15u));
	float *s_d = reinterpret_cast<float *>(s_id + (STAGED ? total : 0u));
	for (uint32_t sh = lane; sh < n_shards; sh += 64)
		nvalid[sh] = 0;
	__syncthreads();
	// one coalesced pass: stage, count the valid cells per shard, check the promise (ascending, valid cells first)
	bool broken = false;
	for (uint32_t i = lane; i < total; i += 64) {
		const uint32_t sh = i / k, j = i - sh * k;
		const float di = in_d[sh * stride_d + first + j];
		const int64_t idi = in_id[sh * stride_id + first + j];
		if (STAGED) {
			s_d[i] = di;
			s_id[i] = idi;
		}
		if (idi >= 0)
			atomicAdd(&nvalid[sh], 1u);
		if (j > 0) {
			const float dp = in_d[sh * stride_d + first + j - 1];
			const int64_t idp = in_id[sh * stride_id + first + j - 1];
			if (idi >= 0 && (idp < 0 || !(dp <= di))) // a valid cell behind an unused one, a descent, or a NaN
				broken = true;
		} else if (idi >= 0 && !(di == di)) {
			broken = true;
		}
	}
	__syncthreads();
	MergeCells<STAGED> cells;
	if (STAGED)
		cells.d = s_d, cells.id = s_id, cells.stride_d = k, cells.stride_id = k;
	else
		cells.d = in_d + first, cells.id = in_id + first, cells.stride_d = stride_d, cells.stride_id = stride_id;
	uint32_t valid = 0;
	for (uint32_t sh = 0; sh < n_shards; ++sh)
		valid += nvalid[sh];
	if (!__ballot(broken)) {
		for (uint32_t i = lane; i < total; i += 64) {
			const uint32_t sh = i / k, j = i - sh * k;
			if (j >= nvalid[sh] || j >= k)
				continue;
			const float di = cells.dist(sh, j);
			const int64_t idi = cells.key(sh, j);
			uint32_t rank = 0;
			for (uint32_t o = 0; o < n_shards && rank < k; ++o)
				rank += merge_cells_before(cells, o, nvalid[o], di, idi);
			if (rank < k) {
				out_d[first + rank] = di;
				out_id[first + rank] = idi;
			}
		}
	} else {
		// lists that are not what the contract promises: rank by counting over every pair (any order, unused cells anywhere)
		for (uint32_t i = lane; i < total; i += 64) {
			const uint32_t sh = i / k, j = i - sh * k;
			const float di = cells.dist(sh, j);
			const int64_t idi = cells.key(sh, j);
			if (idi < 0)
				continue;
			uint32_t rank = 0;
			for (uint32_t o = 0; o < n_shards; ++o)
				for (uint32_t t = 0; t < k; ++t) {
					const float dt = cells.dist(o, t);
					const int64_t idt = cells.key(o, t);
					rank += (idt >= 0 && ((dt < di) || (dt == di && idt < idi))) ? 1u : 0u;
				}
			if (rank < k) {
				out_d[first + rank] = di;
				out_id[first + rank] = idi;
			}
		}
	}
	const uint32_t count = valid < k ? valid : k;
	for (uint32_t i = count + lane; i < k; i += 64) {
		out_d[first + i] = __builtin_inff();
		out_id[first + i] = -1ll;
	}
	if (lane == 0 && out_count)
		out_count[q] = count;
}

#endif // VSS_ENGINE_TU

} // namespace vss
