// wave_distances_bench.hip — cycles per call of the production wave_distances<> on random rows (one wave per block).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I duckdb-vss_amd/csrc wave_distances_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "wave_primitives.h"
using namespace vss;

template <int MT, int NCH, int R>
__global__ __launch_bounds__(64) void k_bench(RowSpace sp, uint32_t n_rows, int n, int iters, unsigned long long *out,
                                              float *sink) {
	__shared__ float4 q[192 * 2];
	__shared__ uint32_t ids[64];
	__shared__ float dist[64];
	const uint32_t lane = threadIdx.x;
	for (uint32_t i = lane; i < sp.V; i += 64)
		q[i] = make_float4(0.01f * i, 0.5f, -0.25f, 1.f);
	uint32_t seed = blockIdx.x * 2654435761u + 12345u + lane * 97u;
	float acc = 0.f;
	unsigned long long total = 0;
	__syncthreads();
	for (int it = 0; it < iters; ++it) {
		seed = seed * 1664525u + 1013904223u;
		if (lane < (uint32_t)n)
			ids[lane] = (seed >> 8) % n_rows;
		__syncthreads();
		const unsigned long long t0 = __builtin_readcyclecounter();
		wave_distances<MT, NCH, R>(sp, q, 1.0f, ids, n, dist);
		const unsigned long long t1 = __builtin_readcyclecounter();
		total += t1 - t0;
		acc += dist[lane % n];
	}
	if (lane == 0)
		out[blockIdx.x] = total;
	if (acc == 123.456f)
		sink[0] = acc;
}

int main() {
	const uint32_t V = 192, n_rows = 300000;
	float4 *d;
	hipMalloc(&d, (size_t)n_rows * V * 16);
	hipMemset(d, 0, (size_t)n_rows * V * 16);
	RowSpace sp{d, V, 64, 6, 1};
	unsigned long long *dc;
	float *sink;
	hipMalloc(&dc, 4096 * 8);
	hipMalloc(&sink, 4);
	for (int grid : {64, 1024}) {
		for (int n : {1, 4, 8, 12, 16, 32}) {
			const int iters = 200;
			hipLaunchKernelGGL((k_bench<1, 3, 8>), dim3(grid), dim3(64), 0, 0, sp, n_rows, n, iters, dc, sink);
			hipDeviceSynchronize();
			std::vector<unsigned long long> h(grid);
			hipMemcpy(h.data(), dc, grid * 8, hipMemcpyDeviceToHost);
			double mean = 0;
			for (auto v : h)
				mean += (double)v / iters;
			printf("cosine grid %4d n=%2d: %8.0f cycles per wave_distances call\n", grid, n, mean / grid);
		}
	}
	return 0;
}
