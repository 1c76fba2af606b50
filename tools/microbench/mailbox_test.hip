// mailbox_test.hip — the search engine's LDS job exchange in isolation (microbenchmark / bring-up check, not product code).
// One workgroup of W waves: wave 0 posts `jobs` jobs (n ids each) through the production Mailbox protocol and waits for
// `done`; the other waves claim chunks with the 64-bit ticket and "score" them (dist[i] = 2 * ids[i] + 1).  Prints
// whether every result was right and the cycles per hand-off.  Bounded spins: cannot hang.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I duckdb-vss_amd/csrc mailbox_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "hnsw_kernels.h"
using namespace vss;

__global__ __launch_bounds__(1024) void k_mailbox(int jobs, int n, uint32_t *out, unsigned long long *cycles) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint32_t *exit_flag = reinterpret_cast<uint32_t *>(smem);
	Mailbox *mb = reinterpret_cast<Mailbox *>(smem + 16);
	uint32_t *ids = reinterpret_cast<uint32_t *>(smem + 256);
	uint32_t *dist = ids + 256;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (threadIdx.x == 0) {
		*exit_flag = 0;
		mb->ticket = 0, mb->done = 0, mb->qa2_bits = 0;
	}
	__syncthreads();
	if (wave != 0) {
		uint32_t polls = 0;
		for (;;) {
			const unsigned long long t = VSS_LDS_LOAD(lds_u64, &mb->ticket);
			if (uniform((int)((uint32_t)t < (uint32_t)(t >> 32)))) {
				for (;;) {
					unsigned long long *scrap = reinterpret_cast<unsigned long long *>(smem + 4096);
					const unsigned long long c64 = VSS_LDS_ADD(lds_u64, lane == 0 ? &mb->ticket : scrap + lane, 16ull);
					const uint32_t c = (uint32_t)uniform((int)(uint32_t)c64), nn = (uint32_t)uniform((int)(uint32_t)(c64 >> 32));
					if (c >= nn)
						break;
					const uint32_t cnt = nn - c < 16 ? nn - c : 16;
					if ((uint32_t)lane < cnt)
						dist[c + lane] = 2 * ids[c + lane] + 1;
					wave_sync();
					VSS_LDS_ADD(lds_u32, lane == 0 ? &mb->done : reinterpret_cast<uint32_t *>(scrap + lane), cnt);
				}
			}
			if (uniform((int)VSS_LDS_LOAD(lds_u32, exit_flag)) || ++polls > (1u << 24))
				return;
			__builtin_amdgcn_s_sleep(1);
		}
	}
	uint32_t bad = 0;
	unsigned long long total = 0;
	for (int j = 0; j < jobs; ++j) {
		for (int i = lane; i < n; i += 64)
			ids[i] = (uint32_t)(j * 131 + i);
		wave_sync();
		const unsigned long long t0 = __builtin_readcyclecounter();
		VSS_LDS_STORE(lds_u32, &mb->done, 0u);
		VSS_LDS_STORE(lds_u64, &mb->ticket, (unsigned long long)(uint32_t)n << 32);
		uint32_t spins = 0;
		while (uniform((int)VSS_LDS_LOAD(lds_u32, &mb->done)) < n && ++spins < (1u << 22))
			__builtin_amdgcn_s_sleep(1);
		wave_sync();
		total += __builtin_readcyclecounter() - t0;
		for (int i = lane; i < n; i += 64)
			bad += dist[i] != 2 * (uint32_t)(j * 131 + i) + 1;
		if (spins >= (1u << 22))
			bad += 1000000;
	}
	for (int o = 32; o >= 1; o >>= 1)
		bad += __shfl_xor(bad, o);
	if (lane == 0) {
		VSS_LDS_STORE(lds_u32, exit_flag, 1u);
		out[blockIdx.x] = bad;
		cycles[blockIdx.x] = total;
	}
}

int main() {
	uint32_t *out;
	unsigned long long *cyc;
	hipMalloc(&out, 4096 * 4);
	hipMalloc(&cyc, 4096 * 8);
	for (int waves : {2, 4, 16}) {
		for (int grid : {1, 256}) {
			for (int n : {1, 16, 40, 200}) {
				const int jobs = 1000;
				hipLaunchKernelGGL(k_mailbox, dim3(grid), dim3(64 * waves), 8192, 0, jobs, n, out, cyc);
				hipError_t e = hipDeviceSynchronize();
				std::vector<uint32_t> h(grid);
				std::vector<unsigned long long> c(grid);
				hipMemcpy(h.data(), out, grid * 4, hipMemcpyDeviceToHost);
				hipMemcpy(c.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
				unsigned long long bad = 0, cy = 0;
				for (int b = 0; b < grid; ++b)
					bad += h[b], cy += c[b];
				printf("waves %2d grid %3d n %3d: %s, wrong results %llu, %.0f cycles per job hand-off round trip\n", waves, grid, n,
				       hipGetErrorString(e), bad, (double)cy / grid / jobs);
				fflush(stdout);
			}
		}
	}
	return 0;
}
