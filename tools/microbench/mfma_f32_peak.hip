// What does the f32 matrix pipe of one MI355X sustain?  (round 5; VERDICT r04 item 5: k_exact_scores_v2 sits at 0.76 of the
// 157.3 TFLOP/s datasheet peak for the third round — how much of the gap is the kernel's and how much the chip's?)
//
// Four loops with the score tile's shape (256-thread workgroups, two per compute unit = two waves per SIMD, four accumulators
// of v_mfma_f32_32x32x2_f32 in rotation, 64 MFMAs per "step"):
//   0  MFMAs only, operands in registers
//   1  + the tile's LDS operand traffic: 16 ds_read_b128 per 64 MFMAs (36-float row stride, as X2_LD)
//   2  + one workgroup barrier per step
//   3  as 0 with v_mfma_f32_16x16x4_f32 (same flops per cycle on paper)
//   4  as 2, but the LDS operands are RANDOM floats (a different value in every cell, full mantissas) instead of sixteen small
//      integers: the same instruction stream, realistic switching activity in the multipliers — what the clock does under it
//   5  as 0 with random register operands
// Prints TFLOP/s per variant; the sustained clock follows from variant 0 (64 flops per cycle and SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_peak mfma_f32_peak.hip && ./mfma_f32_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_peak(float *out, int steps, float seed) {
	__shared__ __attribute__((aligned(16))) float lds[2 * 128 * 36];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	for (int i = tid; i < 2 * 128 * 36; i += 256) {
		uint32_t h = (uint32_t)i * 2654435761u + blockIdx.x * 40503u; // (variant 4: random mantissas, magnitudes around 1)
		h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
		lds[i] = VARIANT == 4 ? __uint_as_float(0x3F000000u | (h & 0x00FFFFFFu)) * ((h >> 31) ? -1.f : 1.f) : seed * (float)(i & 15);
	}
	__syncthreads();
	f32x16 acc[4];
	f32x4 acc4[8]; // (variant 3)
	for (int a = 0; a < 4; ++a)
		for (int e = 0; e < 16; ++e)
			acc[a][e] = 0.f;
	for (int a = 0; a < 8; ++a)
		for (int e = 0; e < 4; ++e)
			acc4[a][e] = 0.f;
	const int a_off = ((wave >> 1) * 64 + (lane & 31)) * 36 + 4 * (lane >> 5);
	const int b_off = 128 * 36 + ((wave & 1) * 64 + (lane & 31)) * 36 + 4 * (lane >> 5);
	float4 av[2], bv[2];
	av[0] = av[1] = bv[0] = bv[1] = make_float4(seed, seed + 1.f, seed + 2.f, seed + 3.f);
	if (VARIANT == 5) {
		uint32_t h = (uint32_t)(blockIdx.x * 256 + tid) * 2654435761u;
		float r[16];
		for (int i = 0; i < 16; ++i) {
			h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
			r[i] = __uint_as_float(0x3F000000u | (h & 0x00FFFFFFu)) * ((h >> 31) ? -1.f : 1.f);
		}
		av[0] = make_float4(r[0], r[1], r[2], r[3]), av[1] = make_float4(r[4], r[5], r[6], r[7]);
		bv[0] = make_float4(r[8], r[9], r[10], r[11]), bv[1] = make_float4(r[12], r[13], r[14], r[15]);
	}
	for (int s = 0; s < steps; ++s) {
#pragma unroll
		for (int g = 0; g < 4; ++g) { // four k-groups of 16 MFMAs, as the tile's step
			if (VARIANT == 1 || VARIANT == 2 || VARIANT == 4) {
#pragma unroll
				for (int i = 0; i < 2; ++i) {
					av[i] = *reinterpret_cast<const float4 *>(lds + a_off + i * 32 * 36 + g * 8);
					bv[i] = *reinterpret_cast<const float4 *>(lds + b_off + i * 32 * 36 + g * 8);
				}
			}
			const float ax[4][2] = {{av[0].x, av[1].x}, {av[0].y, av[1].y}, {av[0].z, av[1].z}, {av[0].w, av[1].w}};
			const float bx[4][2] = {{bv[0].x, bv[1].x}, {bv[0].y, bv[1].y}, {bv[0].z, bv[1].z}, {bv[0].w, bv[1].w}};
#pragma unroll
			for (int c = 0; c < 4; ++c)
#pragma unroll
				for (int i = 0; i < 2; ++i)
#pragma unroll
					for (int j = 0; j < 2; ++j) {
						if (VARIANT == 3) { // 16x16x4: half the flops per instruction, two per slot
							acc4[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[c][i], bx[c][j], acc4[i * 2 + j], 0, 0, 0);
							acc4[4 + i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[c][i], bx[c][j], acc4[4 + i * 2 + j], 0, 0, 0);
						} else {
							acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[c][i], bx[c][j], acc[i * 2 + j], 0, 0, 0);
						}
					}
		}
		if (VARIANT == 2 || VARIANT == 4)
			__syncthreads();
	}
	float sum = 0.f;
	for (int a = 0; a < 4; ++a)
		for (int e = 0; e < 16; ++e)
			sum += acc[a][e];
	for (int a = 0; a < 8; ++a)
		for (int e = 0; e < 4; ++e)
			sum += acc4[a][e];
	if (sum == 12345.678f)
		out[blockIdx.x * 256 + tid] = sum;
}

#define CHECK(x)                                                                                                       \
	do {                                                                                                               \
		hipError_t e = (x);                                                                                            \
		if (e != hipSuccess) {                                                                                         \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                                                      \
			return 1;                                                                                                  \
		}                                                                                                              \
	} while (0)

template <int VARIANT>
static int run(const char *name, float *out, int cus, int steps) {
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	const int grid = 2 * cus * 8; // eight rounds of two workgroups per compute unit
	hipLaunchKernelGGL(k_peak<VARIANT>, dim3(grid), dim3(256), 0, 0, out, steps / 8, 1.0f);
	CHECK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int r = 0; r < 5; ++r) {
		CHECK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL(k_peak<VARIANT>, dim3(grid), dim3(256), 0, 0, out, steps, 1.0f);
		CHECK(hipEventRecord(e1, 0));
		CHECK(hipEventSynchronize(e1));
		float ms = 0;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		best = ms < best ? ms : best;
	}
	// flops: per step and wave 64 MFMAs of 32 x 32 x 2 x 2 flops (variant 3: 128 of 16 x 16 x 4 x 2 — per accumulator half)
	const double per_step = VARIANT == 3 ? 64.0 * 2 * (16.0 * 16 * 4 * 2) : 64.0 * (32.0 * 32 * 2 * 2);
	const double flops = (double)grid * 4 * steps * per_step;
	const double tf = flops / (best * 1e-3) / 1e12;
	printf("%-58s %8.3f ms  %7.1f TFLOP/s = %.3f of 157.3  (implied clock %.2f GHz at 64 flops/cycle/SIMD)\n", name, best, tf, tf / 157.3,
	       tf * 1e12 / (64.0 * 4 * cus) / 1e9);
	return 0;
}

int main() {
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	printf("%s, %d compute units, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
	float *out;
	CHECK(hipMalloc(&out, 2 * cus * 8 * 256 * sizeof(float)));
	const int steps = 2000;
	if (run<0>("0 MFMAs only (32x32x2, four accumulators in rotation)", out, cus, steps)) return 1;
	if (run<1>("1 + 16 ds_read_b128 per 64 MFMAs (the tile's operand reads)", out, cus, steps)) return 1;
	if (run<2>("2 + one workgroup barrier per 64 MFMAs", out, cus, steps)) return 1;
	if (run<3>("3 MFMAs only, 16x16x4", out, cus, steps)) return 1;
	if (run<4>("4 as 2 with RANDOM operands in LDS (real switching activity)", out, cus, steps)) return 1;
	if (run<5>("5 as 0 with random register operands", out, cus, steps)) return 1;
	if (run<0>("0 again (the clock after the runs above)", out, cus, steps)) return 1;
	return 0;
}
