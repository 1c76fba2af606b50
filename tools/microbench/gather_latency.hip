// gather_latency.hip — microbenchmark: what does ONE wave get when it gathers random 3 KiB rows?
// Each wave (one per block) repeatedly picks ROWS random rows, issues NCH x ROWS float4 loads per lane, waits, and
// measures shader cycles per iteration.  Grid sizes 64 / 256 / 1024 / 4096 blocks show the scaling with waves per CU.
// Build: hipcc --offload-arch=gfx950 -O3 gather_latency.hip -o gather_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int ROWS, int NCH>
__global__ __launch_bounds__(64) void k_gather(const float4 *data, uint32_t n_rows, uint32_t V, int iters,
                                               unsigned long long *out_cycles, float *sink) {
	const uint32_t lane = threadIdx.x;
	uint32_t seed = blockIdx.x * 2654435761u + 12345u;
	float acc = 0.f;
	unsigned long long total = 0;
	for (int it = 0; it < iters; ++it) {
		uint32_t rows[ROWS];
#pragma unroll
		for (int r = 0; r < ROWS; ++r) {
			seed = seed * 1664525u + 1013904223u;
			rows[r] = (seed >> 8) % n_rows;
		}
		const unsigned long long t0 = __builtin_readcyclecounter();
		float4 x[NCH][ROWS];
#pragma unroll
		for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
			for (int r = 0; r < ROWS; ++r)
				x[ch][r] = data[(size_t)rows[r] * V + lane + ch * 64];
#pragma unroll
		for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
			for (int r = 0; r < ROWS; ++r)
				acc += x[ch][r].x + x[ch][r].y + x[ch][r].z + x[ch][r].w;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		const unsigned long long t1 = __builtin_readcyclecounter();
		total += t1 - t0;
	}
	if (lane == 0)
		out_cycles[blockIdx.x] = total;
	if (acc == 123.456f)
		sink[0] = acc;
}

template <int ROWS, int NCH>
void run(const float4 *d, uint32_t n_rows, uint32_t V, int grid, const char *label) {
	unsigned long long *dc;
	float *sink;
	hipMalloc(&dc, grid * 8);
	hipMalloc(&sink, 4);
	const int iters = 200;
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	hipLaunchKernelGGL((k_gather<ROWS, NCH>), dim3(grid), dim3(64), 0, 0, d, n_rows, V, 20, dc, sink);
	hipEventRecord(e0);
	hipLaunchKernelGGL((k_gather<ROWS, NCH>), dim3(grid), dim3(64), 0, 0, d, n_rows, V, iters, dc, sink);
	hipEventRecord(e1);
	hipDeviceSynchronize();
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	std::vector<unsigned long long> h(grid);
	hipMemcpy(h.data(), dc, grid * 8, hipMemcpyDeviceToHost);
	double mean = 0;
	for (auto v : h)
		mean += (double)v / iters;
	mean /= grid;
	const double bytes = (double)grid * iters * ROWS * NCH * 1024.0;
	printf("%-8s grid %5d rows/iter %2d x %d KiB: %8.0f cycles/iter/wave, %7.1f GB/s per wave-stream, aggregate %8.1f GB/s\n",
	       label, grid, ROWS, NCH, mean, ROWS * NCH * 1024.0 / (mean / 2.4), bytes / (ms * 1e-3) / 1e9);
	hipFree(dc), hipFree(sink);
}

int main() {
	const uint32_t V = 192; // 768 floats
	for (uint32_t n_rows : {3000u, 1000000u}) {
		float4 *d;
		hipMalloc(&d, (size_t)n_rows * V * 16);
		hipMemset(d, 0, (size_t)n_rows * V * 16);
		const char *label = n_rows == 3000u ? "9MB" : "3GB";
		for (int grid : {64, 256, 1024, 4096}) {
			run<1, 1>(d, n_rows, V, grid, label);
			run<1, 3>(d, n_rows, V, grid, label);
			run<4, 3>(d, n_rows, V, grid, label);
			run<8, 3>(d, n_rows, V, grid, label);
			run<16, 3>(d, n_rows, V, grid, label);
		}
		hipFree(d);
	}
	return 0;
}
