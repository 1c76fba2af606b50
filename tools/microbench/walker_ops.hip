// walker_ops.hip — what do the walker's own operations cost?  (round 6; VERDICT r05 item 1: every regime short of the streaming
// one is bound by the walking wave's instruction stream — the sorted inserts first of all.)
//
// One wave per workgroup, one workgroup per compute unit, no memory traffic beyond LDS: shader-clock cycles of
//   accept   the accept phase of an expansion as level_search_pipelined runs it — n fresh (distance, slot) pairs, one per lane,
//            those that beat the radius inserted one by one — on a FULL list of `limit` entries, with the round-5 list
//            (lane-major: tools/microbench/wave_list_r5.h, incl. its batched merge from six candidates on) and with round 6's
//            blocked, right-aligned list (csrc/wave_primitives.h);
//   pick     the best unexpanded entry: position + distance + slot, then its mark;
//   ahead    the best TWO unexpanded entries' slots (the look-ahead's list requests);
//   gather   n ids through the visited set in LDS (VisitedSet::test_and_set, 32-bit cells) at a given fill.
// Both lists run the SAME pseudo-random candidate stream and must end with identical contents (checked entry by entry on the
// host): the A/B is also a correctness test of the new list against the old one, ties included (TIES = 1: a coarse lattice of
// distances).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I duckdb-vss_amd/csrc -o tools/microbench/walker_ops tools/microbench/walker_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "wave_list_r5.h"

using namespace vss;

// (a cycle-counter read the scheduler cannot move work across)
#define TICK(v)                                                                                                        \
	__builtin_amdgcn_sched_barrier(0);                                                                                 \
	const unsigned long long v = __builtin_readcyclecounter();                                                         \
	__builtin_amdgcn_sched_barrier(0)

struct OpsOut {
	unsigned long long accept_cycles, inserts, pick_cycles, ahead_cycles, expansions;
};

__device__ __forceinline__ uint32_t mix(uint32_t x) {
	x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
	return x;
}

// LIST = WaveList<E> or WaveListR5<E>.  MERGE: the round-5 accept path (merge from six candidates on; needs the staging rows).
template <class LIST, bool MERGE>
__global__ __launch_bounds__(64) void k_accept(int limit, int n_rows, int iters, int accept_one_in, int ties, OpsOut *out,
                                               float *dump_d, uint32_t *dump_s) {
	__shared__ float stage_d[512];
	__shared__ uint32_t stage_s[512];
	__shared__ float row_d[64];
	__shared__ uint32_t row_s[64];
	const int lane = threadIdx.x & 63;
	LIST L;
	L.reset(limit);
	uint32_t seed = 12345u + blockIdx.x * 977u;
	// fill: `limit` entries with distances in (1, 2)
	for (int base = 0; base < limit; base += 64) {
		const uint32_t h = mix(seed + base + lane);
		const float dv = ties ? 1.f + (float)(h % 97u) / 97.f : 1.f + (float)(h >> 8) * (1.f / 16777216.f);
		for (int j = 0; j < 64 && base + j < limit; ++j)
			L.insert(read_lane(dv, j), (uint32_t)(base + j));
	}
	float radius = L.last_distance();
	unsigned long long t_accept = 0, t_pick = 0, t_ahead = 0, n_ins = 0;
	uint32_t next_id = 1000;
	for (int it = 0; it < iters; ++it) {
		// an expansion's fresh scores, as the scoring waves leave them in LDS
		const uint32_t h = mix(seed ^ (uint32_t)(it * 64 + lane) * 2654435761u);
		const float u = (float)(h >> 8) * (1.f / 16777216.f); // [0, 1)
		float dv = radius * (0.25f + u * (float)accept_one_in * 0.75f);
		if (ties)
			dv = __builtin_floorf(dv * 64.f) / 64.f + 1.f / 128.f;
		if (lane < n_rows) {
			row_d[lane] = dv;
			row_s[lane] = next_id + lane;
		}
		next_id += 64;
		lds_sync();
		TICK(t0);
		{
			const bool have = lane < n_rows;
			const float d = have ? row_d[lane] : 0.f;
			const uint32_t id = have ? row_s[lane] : 0;
			unsigned long long pass = __ballot(have && (L.size < limit || d < radius));
			bool merged = false;
			if constexpr (MERGE) {
				if (__popcll(pass) >= 6 && L.merge(d, id, pass, stage_d, stage_s)) {
					radius = L.last_distance();
					n_ins += __popcll(pass);
					merged = true;
				}
			}
			if constexpr (std::is_same<LIST, WaveList<LIST::regs>>::value) { // round 6: the list's own accept loop
				n_ins += __popcll(pass); // (an upper bound: a later candidate may lose to a radius an earlier one shrank)
				L.accept(d, id, pass, radius);
				merged = true;
			}
			if (!merged)
				while (pass) {
					const int j = __builtin_ctzll(pass);
					pass &= pass - 1;
					const float dj = read_lane(d, j);
					if (L.size < limit || dj < radius) {
						L.insert(dj, read_lane(id, j));
						radius = L.last_distance();
						n_ins++;
					}
				}
		}
		TICK(t1);
		// pick: the best unexpanded entry and its mark (every 4th expansion un-marks nothing: the list keeps unexpanded entries
		// because inserts bring new ones)
		float cd = 0.f;
		uint32_t cs = 0;
		int pos;
		if constexpr (std::is_same<LIST, WaveList<LIST::regs>>::value) {
			pos = L.first_unexpanded_entry(cd, cs);
		} else {
			pos = L.first_unexpanded();
			if (pos >= 0)
				L.get(pos, cd, cs);
		}
		if (pos >= 0)
			L.mark_expanded(pos);
		TICK(t2);
		uint32_t s1 = 0, s2 = 0;
		if constexpr (std::is_same<LIST, WaveList<LIST::regs>>::value) {
			L.first_two_unexpanded(s1, s2);
		} else {
			const int a = L.first_unexpanded();
			float xd;
			if (a >= 0) {
				L.get(a, xd, s1);
				const int b = L.next_unexpanded(a);
				if (b >= 0)
					L.get(b, xd, s2);
			}
		}
		TICK(t3);
		seed += (cs ^ s1 ^ s2) & 1u; // (keeps the results alive)
		t_accept += t1 - t0, t_pick += t2 - t1, t_ahead += t3 - t2;
	}
	if (lane == 0) {
		out[blockIdx.x].accept_cycles = t_accept, out[blockIdx.x].inserts = n_ins;
		out[blockIdx.x].pick_cycles = t_pick, out[blockIdx.x].ahead_cycles = t_ahead, out[blockIdx.x].expansions = iters;
	}
	if (blockIdx.x == 0) { // the final list, position by position
		L.dump(stage_d, stage_s);
		lds_sync();
		for (int i = lane; i < L.size; i += 64) {
			dump_d[i] = stage_d[i];
			float xd;
			uint32_t xs;
			(void)xd, (void)xs;
			dump_s[i] = stage_s[i];
		}
		if (lane == 0)
			dump_s[600] = (uint32_t)L.size;
		// expanded marks: position of every unexpanded entry, in order (both lists must agree on them as well)
		int n_un = 0;
		for (int p = L.first_unexpanded(); p >= 0 && n_un < 64; p = L.next_unexpanded(p)) {
			if (lane == 0)
				dump_s[700 + n_un] = (uint32_t)p;
			n_un++;
		}
		if (lane == 0)
			dump_s[699] = (uint32_t)n_un;
	}
}

// n ids through a visited set already holding `fill` keys; `present_pct` per cent of the ids are among those keys (a
// neighbour list mostly names rows the search has seen).  compact_log2 = 0: 2^log2 32-bit cells; else the compact form
// (2^compact_log2 16-bit cells over the same LDS).  (Session C of round 6 built this twice — the engine's compare-and-swap-first
// probes and a read-first variant: profiles/r06c_walker_ops*.txt; the engine now reads first in the 16-bit form only.)
__global__ __launch_bounds__(64) void k_gather(int log2_words, int compact_log2, int fill, int n_ids, int present_pct, int iters,
                                               unsigned long long *out) {
	extern __shared__ uint32_t table[];
	const int lane = threadIdx.x & 63;
	VisitedSet v;
	v.table = table, v.mask = (1u << log2_words) - 1, v.shift = 32 - log2_words, v.limit = 1u << 30, v.count = 0, v.compact = 0;
	if (compact_log2)
		v.mask = (1u << compact_log2) - 1, v.compact = compact_log2;
	unsigned long long total = 0;
	uint32_t fresh = 0;
	for (int it = 0; it < iters; ++it) {
		v.clear();
		uint32_t bad = 0;
		for (int base = 0; base < fill; base += 64)
			if (base + lane < fill)
				v.test_and_set(mix(7u * it + base + lane) % 10000000u, bad);
		wave_sync();
		const bool old_key = (int)(mix(lane * 31u + it) % 100u) < present_pct && fill > 0;
		const uint32_t id = old_key ? mix(7u * it + (mix(lane + it * 131u) % (uint32_t)(fill > 0 ? fill : 1))) % 10000000u
		                            : mix(0x9e3779b9u * (it + 1) + lane) % 10000000u;
		TICK(t0);
		const bool take = lane < n_ids && !v.test_and_set(id, bad);
		const unsigned long long m = __ballot(take);
		lds_sync();
		TICK(t1);
		fresh += __popcll(m) + bad;
		total += t1 - t0;
	}
	if (lane == 0)
		out[blockIdx.x] = total + (fresh == 0xFFFFFFFFu ? 1 : 0);
}

template <class LIST, bool MERGE>
static void run_accept(const char *label, int limit, int n_rows, int one_in, int ties, std::vector<float> &fd, std::vector<uint32_t> &fs) {
	const int grid = 256, iters = 2000;
	OpsOut *d_out;
	float *d_d;
	uint32_t *d_s;
	hipMalloc(&d_out, grid * sizeof(OpsOut));
	hipMalloc(&d_d, 1024 * 4);
	hipMalloc(&d_s, 1024 * 4);
	hipMemset(d_s, 0, 1024 * 4);
	hipLaunchKernelGGL((k_accept<LIST, MERGE>), dim3(grid), dim3(64), 0, 0, limit, n_rows, iters, one_in, ties, d_out, d_d, d_s);
	if (hipDeviceSynchronize() != hipSuccess) {
		printf("%s: kernel failed\n", label);
		exit(1);
	}
	std::vector<OpsOut> o(grid);
	hipMemcpy(o.data(), d_out, grid * sizeof(OpsOut), hipMemcpyDeviceToHost);
	fd.resize(1024), fs.resize(1024);
	hipMemcpy(fd.data(), d_d, 1024 * 4, hipMemcpyDeviceToHost);
	hipMemcpy(fs.data(), d_s, 1024 * 4, hipMemcpyDeviceToHost);
	double acc = 0, ins = 0, pick = 0, ahead = 0, ex = 0;
	for (auto &x : o)
		acc += x.accept_cycles, ins += x.inserts, pick += x.pick_cycles, ahead += x.ahead_cycles, ex += x.expansions;
	printf("%-34s limit %3d rows %2d ties %d: accept %7.0f cycles per expansion (%.2f inserts, %6.0f per insert) | pick+mark %5.0f | "
	       "best two %5.0f\n", label, limit, n_rows, ties, acc / ex, ins / ex, ins ? acc / ins : 0.0, pick / ex, ahead / ex);
	hipFree(d_out), hipFree(d_d), hipFree(d_s);
}

template <int E>
static int ab(int limit, int n_rows, int one_in, int ties) {
	std::vector<float> d0, d1, d2;
	std::vector<uint32_t> s0, s1, s2;
	char label[96];
	snprintf(label, sizeof label, "round 5 list, %d regs, inserts", E);
	run_accept<WaveListR5<E>, false>(label, limit, n_rows, one_in, ties, d0, s0);
	snprintf(label, sizeof label, "round 5 list, %d regs, + merge", E);
	run_accept<WaveListR5<E>, true>(label, limit, n_rows, one_in, ties, d1, s1);
	snprintf(label, sizeof label, "round 6 list, %d regs (blocked)", E);
	run_accept<WaveList<E>, false>(label, limit, n_rows, one_in, ties, d2, s2);
	int bad = 0;
	const uint32_t n = s0[600];
	if (s1[600] != n || s2[600] != n || n != (uint32_t)limit)
		bad++;
	for (uint32_t i = 0; i < n && i < 512; ++i)
		if (std::memcmp(&d0[i], &d2[i], 4) || s0[i] != s2[i] || std::memcmp(&d0[i], &d1[i], 4) || s0[i] != s1[i])
			bad++;
	for (uint32_t i = 699; i < 700 + (s0[699] < 64 ? s0[699] : 64); ++i)
		if (s0[i] != s2[i] || s0[i] != s1[i])
			bad++;
	printf("    final lists (%u entries, %u unexpanded) identical across the three: %s\n", n, s0[699], bad ? "NO" : "yes");
	return bad;
}

int main() {
	int bad = 0;
	// (rows per expansion, one in N passes the radius) ~ the regimes of DESIGN §4.2: M0 64 / ef 60, M0 32 / ef 64, M0 32 / ef 512
	for (int ties = 0; ties <= 1; ++ties) {
		bad += ab<1>(64, 20, 6, ties);
		bad += ab<2>(60, 25, 8, ties);
		bad += ab<2>(128, 25, 8, ties);
		bad += ab<4>(256, 12, 3, ties);
		bad += ab<8>(512, 8, 2, ties);
		bad += ab<8>(480, 24, 2, ties);
	}
	// the visited set
	const char *how = "as the engine probes it";
	unsigned long long *d_c;
	hipMalloc(&d_c, 256 * 8);
	for (int compact : {0, 14})
		for (int fill_pct : {5, 15, 30, 50})
			for (int n : {32, 64}) {
				const int log2w = 13, cells = compact ? (1 << compact) : (1 << log2w);
				const int fill = cells * fill_pct / 100, iters = 200;
				hipLaunchKernelGGL(k_gather, dim3(256), dim3(64), (1u << log2w) * 4, 0, log2w, compact, fill, n, 70, iters, d_c);
				if (hipDeviceSynchronize() != hipSuccess) {
					printf("gather kernel failed\n");
					return 1;
				}
				std::vector<unsigned long long> c(256);
				hipMemcpy(c.data(), d_c, 256 * 8, hipMemcpyDeviceToHost);
				double t = 0;
				for (auto x : c)
					t += x;
				printf("visited set (%s), %s, %2d %% full: %2d ids (70 %% seen before): %6.0f cycles\n",
				       compact ? "16-bit cells x 16384" : "32-bit cells x 8192", how, fill_pct, n, t / 256 / iters);
			}
	printf(bad ? "MISMATCH\n" : "all lists identical\n");
	return bad ? 2 : 0;
}
