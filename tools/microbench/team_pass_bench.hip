// team_pass_bench.hip — what does ONE scoring pass of a search team cost?  (microbenchmark, not product code)
// A workgroup of W waves mimics k_search's team_distances: the walking wave publishes n random row ids in LDS, a
// workgroup barrier, every wave scores its slice with the production wave_distances<>, a second barrier.  Cycles per
// pass are measured on the walking wave for n = 4 .. 64 rows, on an idle chip (64 workgroups), one workgroup per CU
// (256) and four per CU (1024).  Rows are 768 floats (cosine) out of a 3 GB / 30 GB table.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I duckdb-vss_amd/csrc team_pass_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "wave_primitives.h"
using namespace vss;

template <int MT, int NCH, int R, int W>
__global__ __launch_bounds__(64 * W) void k_team(RowSpace sp, uint32_t n_rows, int n, int iters, unsigned long long *out,
                                                 float *sink) {
	__shared__ float4 q[384];
	__shared__ uint32_t ids[128];
	__shared__ float dist[128];
	const uint32_t lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	for (uint32_t i = threadIdx.x; i < sp.V; i += 64 * W)
		q[i] = make_float4(0.01f * i, 0.5f, -0.25f, 1.f);
	uint32_t seed = blockIdx.x * 2654435761u + 12345u + lane * 97u;
	float acc = 0.f;
	unsigned long long total = 0, t_mid = 0;
	__syncthreads();
	const int pass = R * (64 >> sp.logG);
	for (int it = 0; it < iters; ++it) {
		if (wave == 0) {
			seed = seed * 1664525u + 1013904223u;
			for (int i = lane; i < n; i += 64)
				ids[i] = ((seed >> 8) + 7919u * i) % n_rows;
		}
		const unsigned long long t0 = __builtin_readcyclecounter();
		__syncthreads();
		const unsigned long long t1 = __builtin_readcyclecounter();
		for (int off = wave * pass; off < n; off += W * pass)
			wave_distances<MT, NCH, R>(sp, q, 1.0f, ids + off, n - off < pass ? n - off : pass, dist + off);
		const unsigned long long t2 = __builtin_readcyclecounter();
		__syncthreads();
		const unsigned long long t3 = __builtin_readcyclecounter();
		total += t3 - t0;
		t_mid += t2 - t1;
		acc += dist[lane % n];
	}
	if (threadIdx.x == 0) {
		out[2 * blockIdx.x] = total;
		out[2 * blockIdx.x + 1] = t_mid;
	}
	if (acc == 123.456f)
		sink[0] = acc;
}

template <int R, int W>
void run(const RowSpace &sp, uint32_t n_rows, const char *label, unsigned long long *dc, float *sink) {
	for (int grid : {64, 256, 1024}) {
		for (int n : {1, 4, 8, 16, 21, 32, 64}) {
			const int iters = 100;
			hipLaunchKernelGGL((k_team<1, 3, R, W>), dim3(grid), dim3(64 * W), 0, 0, sp, n_rows, n, 10, dc, sink);
			hipEvent_t e0, e1;
			hipEventCreate(&e0), hipEventCreate(&e1);
			hipEventRecord(e0);
			hipLaunchKernelGGL((k_team<1, 3, R, W>), dim3(grid), dim3(64 * W), 0, 0, sp, n_rows, n, iters, dc, sink);
			hipEventRecord(e1);
			hipDeviceSynchronize();
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			std::vector<unsigned long long> h(2 * grid);
			hipMemcpy(h.data(), dc, 2 * grid * 8, hipMemcpyDeviceToHost);
			double mean = 0, mid = 0;
			for (int b = 0; b < grid; ++b)
				mean += (double)h[2 * b] / iters, mid += (double)h[2 * b + 1] / iters;
			printf("%s W=%d R=%d grid %4d n=%2d: %7.0f cycles per pass (walker's own slice %7.0f), %7.1f GB/s aggregate\n", label,
			       W, R, grid, n, mean / grid, mid / grid, (double)grid * iters * n * 3072.0 / (ms * 1e-3) / 1e9);
			hipEventDestroy(e0), hipEventDestroy(e1);
		}
	}
}

int main(int argc, char **argv) {
	const uint32_t V = 192;
	const uint32_t n_rows = argc > 1 ? (uint32_t)atoll(argv[1]) : 1000000u;
	float4 *d;
	if (hipMalloc(&d, (size_t)n_rows * V * 16) != hipSuccess)
		return 1;
	hipMemset(d, 0, (size_t)n_rows * V * 16);
	RowSpace sp {d, V, 64, 6, 1};
	unsigned long long *dc;
	float *sink;
	hipMalloc(&dc, 2 * 4096 * 8);
	hipMalloc(&sink, 4);
	char label[32];
	snprintf(label, sizeof label, "%.1fGB", (double)n_rows * V * 16 / 1e9);
	run<4, 4>(sp, n_rows, label, dc, sink);
	run<8, 1>(sp, n_rows, label, dc, sink);
	run<4, 8>(sp, n_rows, label, dc, sink);
	run<8, 4>(sp, n_rows, label, dc, sink);
	return 0;
}
