// vss_engine.hip — host side of libvssgpu.so: device memory management, the batch-synchronous build schedule,
// kernel launches, the usearch-compatible stream format, and the C ABI declared in include/vssgpu.h.
//
// What it replaces in the reference: the `unum::usearch::index_dense_gt<row_t> index` member of HNSWIndex
// (reference src/include/hnsw/hnsw_index.hpp:45) — i.e. index_dense.hpp + index.hpp + index_plugins.hpp of the
// vendored usearch.  Nothing here depends on oracle/ (test infrastructure); without a HIP device every entry
// point fails loudly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#define VSS_ENGINE_TU
#include "../../include/vssgpu.h"
#include "launchers.h"
#include "host_logic.h"

using namespace vss;
using namespace vss::host;

// the constants host_logic.h restates for the CPU tests are the kernels'
static_assert(host::TOUCH_LISTS_BIT == TOUCH_LISTS, "ListTouch bit");
static_assert(host::SearchShapePolicy().team_box_bytes == TEAM_BOX_BYTES, "team box size");
static_assert(team_touch_max_lines(0) == 8 && team_touch_max_lines(1) == 8 && team_touch_max_lines(3) == 24, "helper touch window");

namespace {

// ---------------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------------
struct HipError {
	hipError_t code;
	const char *what;
};

#define HIP_TRY(expr)                                                                                                  \
	do {                                                                                                               \
		hipError_t _e = (expr);                                                                                        \
		if (_e != hipSuccess)                                                                                          \
			throw HipError {_e, #expr};                                                                                \
	} while (0)

template <typename T>
struct DevBuf {
	T *p = nullptr;
	size_t n = 0;
	void free() {
		if (p)
			(void)hipFree(p);
		p = nullptr;
		n = 0;
	}
	// grow to at least `want` elements, preserving the first `keep` elements
	void ensure(size_t want, size_t keep, hipStream_t s, int fill_byte = -1) {
		if (want <= n)
			return;
		T *np = nullptr;
		HIP_TRY(hipMalloc(&np, want * sizeof(T)));
		if (fill_byte >= 0)
			HIP_TRY(hipMemsetAsync(np, fill_byte, want * sizeof(T), s));
		if (p && keep)
			HIP_TRY(hipMemcpyAsync(np, p, keep * sizeof(T), hipMemcpyDeviceToDevice, s));
		HIP_TRY(hipStreamSynchronize(s));
		if (p)
			(void)hipFree(p);
		p = np;
		n = want;
	}
	// grow to at least `want` elements, contents undefined; false (nothing changed but a freed buffer) when the device has no room
	bool try_ensure(size_t want) {
		if (want <= n)
			return true;
		free();
		T *np = nullptr;
		if (hipMalloc(&np, want * sizeof(T)) != hipSuccess) {
			(void)hipGetLastError();
			return false;
		}
		p = np;
		n = want;
		return true;
	}
};

} // namespace

// ---------------------------------------------------------------------------------------------------------
// the index
// ---------------------------------------------------------------------------------------------------------
// result.error.what() of the calling thread's last failed call (searches run concurrently: one string per thread)
static thread_local std::string tls_error;

struct vss_index {
	// Reader/writer discipline of the reference (SURVEY §8b "Threading"): any number of concurrent searches (shared), every
	// call that changes the graph or its buffers exclusive.  Searches lease one of the pooled contexts below.
	std::shared_mutex rw;
	std::mutex stats_mu; // last_stats / last_query_stats / timing[0]
	int device = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;

	// configuration (reference hnsw_index.cpp:181-217)
	uint64_t dim = 0, M = 16, M0 = 32, efc = 128, efs = 64;
	int metric = 0;
	uint32_t V = 0, G = 1, logG = 0;
	uint64_t max_batch = 32768, growth_div = 32; // (round 3: cap 16384 -> 32768: fewer, larger phase-A launches, +5 % rows/s at 10M rows, recall unchanged)

	// host-side graph bookkeeping
	uint64_t limit_members = 0, limit_threads = 0;
	uint64_t capacity = 0;
	uint64_t count = 0;  // linked nodes (incl. tombstones)
	uint64_t staged = 0; // staged but not yet linked: slots [count, count + staged)
	int max_level = -1;
	uint32_t entry = 0;
	double inv_log_m = 0;
	LevelRng rng;
	std::vector<uint8_t> levels_h;
	std::vector<uint32_t> upper_off_h;
	std::vector<int64_t> keys_h;
	std::vector<uint32_t> list_owner_h;
	uint64_t n_upper = 0;
	uint64_t tombstones = 0;
	// some list may name a slot twice: a slot was re-used while stale links still pointed at it (or a loaded stream holds
	// such a list).  Decides how a chunk of a list goes through the visited set (mark_first_visit).
	bool lists_may_repeat = false;
	KeyMap keymap;
	FreeRing free_slots;
	std::atomic<uint64_t> progress_linked {0}, progress_total {0}; // vss_build_progress (read without the index lock)

	// Staged rows that take over a tombstoned slot (the reference's update() path, index_dense.hpp:1766-1793): their
	// new vectors wait in d_pending until the node has been re-linked, because every distance to the slot taken during
	// that re-link still reads the OLD vector (index.hpp:2801-2859).  Once a reuse row is staged the row order of the
	// staged set is explicit: st_slot[i] = slot of staged row i, st_src[i] = its row in d_pending or EMPTY_SLOT.
	std::vector<uint32_t> st_slot, st_src;
	std::vector<int64_t> pending_keys;
	uint64_t n_pending = 0;
	DevBuf<float> d_pending;
	DevBuf<uint32_t> d_row_slot, d_row_src, d_parked;

	// device-side graph
	DevBuf<float> d_vectors;
	DevBuf<uint32_t> d_links0, d_links_up, d_upper_off, d_list_owner;
	DevBuf<uint8_t> d_levels;
	DevBuf<int64_t> d_keys;

	// build scratch
	DevBuf<uint32_t> d_req_list, d_req_src, d_req_rank, d_sorted_src, d_touched, d_list_count, d_list_offset,
	    d_counters;
	DevBuf<float> d_req_d, d_sorted_d;
	uint32_t *h_counters = nullptr; // pinned
	DevBuf<uint32_t> d_node_status, d_work_build;
	DevBuf<float> d_build_list; // candidate lists in HBM (ef_construction > 512)
	uint32_t *h_debug = nullptr; // debug builds (-DVSS_PARANOID): 64 words of pinned host memory the kernels leave notes in
	DevBuf<unsigned long long> d_work_stats; // cumulative {phase A distances, phase A expansions, phase B distances}
	std::vector<uint32_t> h_node_status;

	// search contexts: independent in-flight batched probes over the same (read-only) graph — the analogue of usearch's
	// per-thread search contexts (index.hpp:2213-2240, leased in index_dense.hpp:1730-1745)
	struct SearchCtx {
		hipStream_t stream = nullptr;
		bool own_stream = false;
		DevBuf<uint32_t> d_stats, d_status, d_work, d_global_hash, d_retry_hash, d_queue;
		DevBuf<float> d_list_buf, d_cand_buf;
		uint32_t cand_cap = 0;
		// staging of the host-pointer entry points (one set per context, so that concurrent callers never share any)
		DevBuf<float> d_q, d_out_d;
		DevBuf<int64_t> d_out_keys;
		DevBuf<uint32_t> d_out_count;
		DevBuf<uint64_t> d_filter;
		unsigned char *pinned_io = nullptr;
		size_t pinned_cap = 0;
		bool leased = false;
		DevBuf<unsigned long long> d_phase;
		uint32_t *h_status = nullptr, *h_stats = nullptr; // pinned
		uint32_t *h_queue = nullptr; // pinned, written by the kernel: [1] engine error, [2] "last query handed out",
		                             // [3] queries answered (the flag wait of the one-query probe)
		bool flag_wait = false;      // this launch is waited for on h_queue[3], not on the stream; no events around it
		bool unsynced = false;       // a flag wait returned: the kernel's last instructions may still be in flight
		uint32_t flag_target = 0;    // h_queue[3] once this launch has answered all its queries (grows along a chain of launches)
		std::chrono::steady_clock::time_point flag_t0;
		uint32_t queue_sel = 0;      // which of the two query counters in d_queue the next launch takes
		size_t h_cap = 0;
		hipEvent_t ev0 = nullptr, ev1 = nullptr;
		bool pending = false;
		bool direct_io = false; // the kernel writes status / counters straight into the pinned host arrays
		SearchArgs args;
		uint64_t nq = 0, limit = 0, list_cap = 0;
		uint32_t bump = 0;
		uint32_t min_hash_log2 = 0; // after a visited-set overflow: the next pass's table is at least this large
		double kernel_ms = 0;
		uint64_t stats[4] = {0, 0, 0, 0};
	};
	// contexts 0 .. EXPLICIT_CTX-1 belong to the caller (vss_search_batch_device_begin/_end; 0 is also the blocking
	// device-pointer calls' context and runs on the index stream); the rest are leased by the host-pointer entry points
	static constexpr int EXPLICIT_CTX = 4, MAX_CTX = 12;
	SearchCtx ctx[MAX_CTX];
	std::mutex lease_mu;
	std::condition_variable lease_cv;
	int lease_context() {
		std::unique_lock<std::mutex> lk(lease_mu);
		for (;;) {
			for (int i = EXPLICIT_CTX; i != MAX_CTX; ++i)
				if (!ctx[i].leased) {
					ctx[i].leased = true;
					return i;
				}
			lease_cv.wait(lk);
		}
	}
	void return_context(int i) {
		{
			std::lock_guard<std::mutex> lk(lease_mu);
			ctx[i].leased = false;
		}
		lease_cv.notify_one();
	}
	std::mutex ctx0_mu;  // blocking device-pointer searches share context 0
	std::mutex exact_mu; // the exact path's scratch (score tiles, norms) is shared: exact searches take turns

	// search scratch
	uint64_t last_stats[4] = {0, 0, 0, 0};
	std::vector<uint32_t> last_query_stats;
	// kernel timing (hipEvents on the index's stream): [0] last search kernel(s) ms, [1] build phase A ms,
	// [2] build phase B (count/alloc/scatter/link) ms, [3] build host wall ms, [4] build batches, [5] build retries
	double timing[6] = {0, 0, 0, 0, 0, 0};
	hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
	void ensure_events() {
		for (auto &e : ev)
			if (!e)
				HIP_TRY(hipEventCreate(&e));
	}

	// exact-search scratch
	DevBuf<float> d_row_norm2, d_q_norm2, d_scores, d_best_s, d_qpad, d_cand_s;
	DevBuf<uint32_t> d_best_i, d_cand_cnt, d_cand_i;
	bool exact_filter = true; // the select folded into the score tile's epilogue from the second chunk on (VSS_EXACT_FILTER=0: A/B)
	uint64_t norms_valid_for = ~0ull; // value of `mutations` the norms were computed at
	uint64_t mutations = 0;

	int fail(const char *fmt, ...) {
		char buf[512];
		va_list ap;
		va_start(ap, fmt);
		vsnprintf(buf, sizeof buf, fmt, ap);
		va_end(ap);
		tls_error = buf;
		return VSS_ERROR;
	}

	uint32_t list_cap_max() const {
		return (uint32_t)((std::max(M, M0) + 63) / 64 * 64);
	}
	// limit of the insert search = config.expansion, as connect_node_across_levels_ passes it (usearch index.hpp:3648);
	// the max(max(M0, M) + 1, expansion) of index.hpp:2712-2713 is only the reservation of `top`
	uint32_t top_limit() const {
		return (uint32_t)std::max<uint64_t>(1, efc);
	}

	GraphView view() const {
		GraphView gv;
		gv.sp.vectors = reinterpret_cast<const float4 *>(d_vectors.p);
		gv.sp.V = V;
		gv.sp.G = G;
		gv.sp.logG = logG;
		gv.sp.metric = metric;
		gv.sp.debug_rows = (uint32_t)(count + staged);
		gv.sp.debug = h_debug;
		gv.dim = (uint32_t)dim;
		gv.M = (uint32_t)M;
		gv.M0 = (uint32_t)M0;
		gv.links0 = d_links0.p;
		gv.links_up = d_links_up.p;
		gv.upper_off = d_upper_off.p;
		gv.keys = d_keys.p;
		gv.list_id_base = (uint32_t)capacity;
		gv.filter = nullptr;
		gv.filter_bits = 0;
		gv.twins = lists_may_repeat ? 1u : 0u;
		return gv;
	}

	void release() {
		d_vectors.free(), d_links0.free(), d_links_up.free(), d_upper_off.free(), d_list_owner.free();
		d_levels.free(), d_keys.free();
		d_req_list.free(), d_req_src.free(), d_req_rank.free(), d_sorted_src.free(), d_touched.free();
		d_list_count.free(), d_list_offset.free(), d_counters.free(), d_req_d.free(), d_sorted_d.free();
		d_node_status.free(), d_work_build.free(), d_work_stats.free(), d_build_list.free();
		if (h_debug)
			(void)hipHostFree(h_debug);
		h_debug = nullptr;
		d_pending.free(), d_row_slot.free(), d_row_src.free(), d_parked.free();
		d_global_hash.free(), d_row_norm2.free(), d_q_norm2.free(), d_scores.free(), d_best_s.free(), d_qpad.free();
		d_best_i.free(), d_cand_s.free(), d_cand_cnt.free(), d_cand_i.free();
		if (h_counters)
			(void)hipHostFree(h_counters);
		h_counters = nullptr;
		for (auto &e : ev) {
			if (e)
				(void)hipEventDestroy(e);
			e = nullptr;
		}
		for (auto &c : ctx) {
			if (c.unsynced && c.stream)
				(void)hipStreamSynchronize(c.stream);
			c.d_stats.free(), c.d_status.free(), c.d_work.free(), c.d_global_hash.free(), c.d_retry_hash.free(), c.d_phase.free();
			c.d_queue.free(), c.d_list_buf.free(), c.d_cand_buf.free();
			c.d_q.free(), c.d_out_d.free(), c.d_out_keys.free(), c.d_out_count.free(), c.d_filter.free();
			if (c.pinned_io)
				(void)hipHostFree(c.pinned_io);
			if (c.h_status)
				(void)hipHostFree(c.h_status), (void)hipHostFree(c.h_stats);
			if (c.h_queue)
				(void)hipHostFree(c.h_queue);
			if (c.ev0)
				(void)hipEventDestroy(c.ev0), (void)hipEventDestroy(c.ev1);
			if (c.own_stream && c.stream)
				(void)hipStreamDestroy(c.stream);
			c = SearchCtx();
		}
	}

	void reset_graph() {
		limit_members = limit_threads = capacity = count = staged = 0;
		max_level = -1;
		entry = 0;
		rng = LevelRng();
		levels_h.clear(), upper_off_h.clear(), keys_h.clear(), list_owner_h.clear();
		n_upper = tombstones = 0;
		lists_may_repeat = false;
		keymap = KeyMap();
		free_slots = FreeRing();
		st_slot.clear(), st_src.clear(), pending_keys.clear();
		n_pending = 0;
		mutations++;
	}

	// Mutating calls free or rewrite the arrays a search kernel of an unfinished begin/end probe may still be reading (its
	// stream is not the index stream): they are refused until every context has been completed with vss_search_batch_end.
	int refuse_while_probing(const char *what) {
		for (int i = 0; i != MAX_CTX; ++i)
			if (ctx[i].pending)
				return fail("%s while search context %d still has a batch in flight (call vss_search_batch_end first)", what, i);
		return VSS_OK;
	}

	// ------------------------------------------------------------------ reserve
	void ensure_upper(uint64_t lists) {
		if (lists <= d_links_up.n / std::max<uint64_t>(1, M))
			return;
		uint64_t want = std::max<uint64_t>(lists + lists / 2, 1024);
		d_links_up.ensure(want * M, n_upper * M, stream, 0xFF);
		d_list_owner.ensure(want, n_upper, stream);
	}

	int reserve(uint64_t members, uint64_t threads) {
		if (threads <= limit_threads && members <= limit_members)
			return VSS_OK;
		if (refuse_while_probing("vss_reserve") != VSS_OK)
			return VSS_ERROR;
		if (members >= 0x7FFFFFFFull)
			return fail("capacity above 2^31-1 slots is not supported");
		if (members < count + staged)
			return fail("cannot shrink below the number of stored vectors");
		limit_members = members;
		limit_threads = threads;
		const uint64_t live = count + staged;
		const uint64_t stride = (uint64_t)V * 4;
		d_vectors.ensure(members * stride, live * stride, stream, 0);
		d_links0.ensure(members * M0, live * M0, stream, 0xFF);
		d_upper_off.ensure(members, live, stream, 0);
		d_levels.ensure(members, live, stream, 0);
		d_keys.ensure(members, live, stream, 0);
		ensure_upper((uint64_t)(members / std::max<double>(1.0, (double)M - 1.0) * 1.25) + 1024);
		capacity = members;
		levels_h.resize(members, 0);
		upper_off_h.resize(members, 0);
		keys_h.resize(members, 0);
		rng = LevelRng(); // usearch replaces every thread context (and its generator) when it grows
		mutations++;
		return VSS_OK;
	}

	// ------------------------------------------------------------------ staging
	// Assign slots / levels for `n` valid rows whose keys are already in keys_h[first..], then upload metadata.
	int stage_metadata(uint64_t first, uint64_t n) {
		uint64_t upper_needed = n_upper;
		for (uint64_t i = 0; i != n; ++i) {
			const uint8_t lv = rng.stored_level(inv_log_m);
			levels_h[first + i] = lv;
			upper_off_h[first + i] = (uint32_t)upper_needed;
			upper_needed += lv;
		}
		ensure_upper(upper_needed);
		list_owner_h.resize(upper_needed);
		for (uint64_t i = 0; i != n; ++i)
			for (int l = 0; l < levels_h[first + i]; ++l)
				list_owner_h[upper_off_h[first + i] + l] = (uint32_t)(first + i);
		if (upper_needed > n_upper) {
			HIP_TRY(hipMemcpyAsync(d_list_owner.p + n_upper, list_owner_h.data() + n_upper,
			                       (upper_needed - n_upper) * 4, hipMemcpyHostToDevice, stream));
			HIP_TRY(hipMemsetAsync(d_links_up.p + n_upper * M, 0xFF, (upper_needed - n_upper) * M * 4, stream));
		}
		n_upper = upper_needed;
		HIP_TRY(hipMemcpyAsync(d_levels.p + first, levels_h.data() + first, n, hipMemcpyHostToDevice, stream));
		HIP_TRY(hipMemcpyAsync(d_upper_off.p + first, upper_off_h.data() + first, n * 4, hipMemcpyHostToDevice,
		                       stream));
		HIP_TRY(hipMemcpyAsync(d_keys.p + first, keys_h.data() + first, n * 8, hipMemcpyHostToDevice, stream));
		HIP_TRY(hipMemsetAsync(d_links0.p + first * M0, 0xFF, n * M0 * 4, stream));
		if (keymap.ready)
			for (uint64_t i = 0; i != n; ++i)
				keymap.put(keys_h[first + i], (uint32_t)(first + i));
		staged += n;
		mutations++;
		return VSS_OK;
	}

	int stage(const int64_t *rowids, const float *vecs, const uint64_t *validity, uint64_t n, bool device_ptrs) {
		if (!n)
			return VSS_OK;
		if (refuse_while_probing("vss_stage_batch") != VSS_OK)
			return VSS_ERROR;
		// which rows are valid (DuckDB validity mask: bit set = valid)
		const bool sparse = validity && !device_ptrs;
		std::vector<uint64_t> rows;
		if (sparse) {
			rows.reserve(n);
			for (uint64_t i = 0; i != n; ++i)
				if (validity[i >> 6] & (1ull << (i & 63)))
					rows.push_back(i);
		}
		const uint64_t nv = sparse ? rows.size() : n;
		if (!nv)
			return VSS_OK;
		// the keys of the valid rows, on the host
		std::vector<int64_t> kbuf(nv);
		if (device_ptrs) {
			HIP_TRY(hipMemcpyAsync(kbuf.data(), rowids, n * 8, hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipStreamSynchronize(stream));
		} else if (!sparse) {
			std::memcpy(kbuf.data(), rowids, n * 8);
		} else {
			for (uint64_t i = 0; i != nv; ++i)
				kbuf[i] = rowids[rows[i]];
		}
		if (keymap.ready) { // duplicate check as usearch index_dense.hpp:1752-1753 (only when the map exists)
			uint32_t s;
			for (uint64_t i = 0; i != nv; ++i)
				if (keymap.find(kbuf[i], s))
					return fail("Duplicate keys not allowed in high-level wrappers");
		}
		// Slots: every add() first asks the free ring (index_dense.hpp:1767-1771), so the leading rows of the chunk take
		// over tombstoned slots in ring order until it runs dry; the rest are appended.
		std::vector<uint32_t> reused;
		{
			FreeRing before = free_slots;
			uint32_t s;
			while (reused.size() < nv && free_slots.try_pop(s))
				reused.push_back(s);
			if (count + staged + (nv - reused.size()) > capacity) { // usearch index.hpp:2728-2731
				free_slots = before;
				return fail("Reserve capacity ahead of insertions!");
			}
		}
		const uint64_t nr = reused.size(), first = count + staged;
		const size_t stride = (size_t)V * 4;
		const hipMemcpyKind kind = device_ptrs ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
		// copy the valid rows [a, b) of the chunk to consecutive padded rows at dst
		auto copy_rows = [&](uint64_t a, uint64_t b, float *dst) {
			if (a == b)
				return;
			if (!sparse) {
				HIP_TRY(hipMemcpy2DAsync(dst, stride * 4, vecs + a * dim, dim * 4, dim * 4, b - a, kind, stream));
				return;
			}
			for (uint64_t i = a; i < b;) { // runs of consecutive valid rows
				uint64_t j = i;
				while (j + 1 < b && rows[j + 1] == rows[j] + 1)
					j++;
				HIP_TRY(hipMemcpy2DAsync(dst + (i - a) * stride, stride * 4, vecs + rows[i] * dim, dim * 4, dim * 4,
				                         j - i + 1, kind, stream));
				i = j + 1;
			}
		};
		if (nr) {
			if (st_slot.empty()) // rows staged so far were all appended: make their order explicit
				for (uint64_t i = 0; i != staged; ++i)
					st_slot.push_back((uint32_t)(count + i)), st_src.push_back(EMPTY_SLOT);
			d_pending.ensure((n_pending + nr) * stride, n_pending * stride, stream, 0);
			copy_rows(0, nr, d_pending.p + n_pending * stride);
			for (uint64_t i = 0; i != nr; ++i) {
				st_slot.push_back(reused[i]), st_src.push_back((uint32_t)(n_pending + i));
				pending_keys.push_back(kbuf[i]);
				keys_h[reused[i]] = kbuf[i]; // the device copy stays FREE (invisible to searches) until the re-link
				if (keymap.ready)
					keymap.put(kbuf[i], reused[i]);
			}
			n_pending += nr;
			mutations++; // `tombstones` drops only when build_finalize has re-linked the slot and published its key on the device
		}
		int rc = VSS_OK;
		if (nv > nr) {
			copy_rows(nr, nv, d_vectors.p + first * stride);
			std::memcpy(keys_h.data() + first, kbuf.data() + nr, (nv - nr) * 8);
			if (!st_slot.empty())
				for (uint64_t i = 0; i != nv - nr; ++i)
					st_slot.push_back((uint32_t)(first + i)), st_src.push_back(EMPTY_SLOT);
			rc = stage_metadata(first, nv - nr);
		}
		if (!device_ptrs)
			HIP_TRY(hipStreamSynchronize(stream)); // the caller may recycle its chunk now
		return rc;
	}

	LaunchCfg launch_cfg(uint32_t grid, uint32_t lds, uint64_t list_limit) const {
		LaunchCfg c;
		c.nch = (V % G == 0 && !force_looping) ? V / G : 0; // chunks per lane when the row fills every lane evenly, else the looping kernels
		c.regs = (uint32_t)((list_limit + 63) / 64);
		c.grid = grid;
		c.lds = lds;
		c.threads = 64;
		c.stream = stream;
		return c;
	}
	template <typename A>
	void launch_by_metric(hipError_t (*f0)(const A &, const LaunchCfg &), hipError_t (*f1)(const A &, const LaunchCfg &),
	                      hipError_t (*f2)(const A &, const LaunchCfg &), const A &a, const LaunchCfg &c) const {
		HIP_TRY((metric == 0 ? f0 : metric == 1 ? f1 : f2)(a, c));
	}

	// Visited-set capacity: a level search touches roughly 16-35 x limit nodes (more with wide level-0 lists); the
	// table must stay below 7/8 full.  Tables above HASH_LDS_MAX_LOG2 live in HBM (see carve_lds).
	// Searches keep tables up to 32 KiB in LDS (measured faster at ef <= 128); the build keeps only <= 8 KiB there: with
	// the table in HBM/L2 phase A runs 8 instead of 3 waves per CU and is 1.6x faster (2M x 768, M=32).
	// The search engine (k_search): one persistent workgroup of `search_waves` waves per compute unit, the first
	// `search_walkers` of them walking one query each (0 = chosen per launch from the batch size), the rest scoring.
	// VSS_SEARCH_WAVES / VSS_SEARCH_WALKERS in the environment override them (A/B measurements).
	uint32_t search_waves = 16, search_walkers = 0;
	// walkers per workgroup the automatic choice may use (the kernel admits ENGINE_MAX_WALKERS = 8; VSS_SEARCH_WALKERS_CAP)
	uint32_t search_walkers_cap = 4;
	uint32_t search_wgs_per_cu = 1; // engine workgroups per compute unit (VSS_SEARCH_WGS_PER_CU; only with fewer than 16 waves each)
	// look-ahead while <= this many walkers of a workgroup still run (VSS_SEARCH_SPEC / vss_set_search_lookahead).  OFF by
	// default: bit-identical results, but measured slower — the probe sits on the walker's critical path
	uint32_t search_spec_active = 0;
	// The solo shape (k_search_solo: one wave per query scoring its own rows) answers launches of FEW queries over NARROW
	// rows — above all the single-query probe of HNSW_INDEX_SCAN: 0 = never, 1 = automatic (at most solo_max_queries
	// queries and a whole level-0 list of rows within solo_max_bytes: one wave pulls that about as fast as it could be
	// spread over scoring waves, without the exchange), 2 = always.  VSS_SEARCH_SOLO / VSS_SEARCH_SOLO_MAX override.
	uint32_t search_solo = 1, solo_max_queries = 32, solo_max_bytes = 32 * 1024;
	// The choice itself is pure host logic (host_logic.h: wants_solo / choose_search_shape, CPU-tested): narrow rows — one
	// wave pulls a whole level-0 list about as fast as it could be spread over scoring waves; as teams (helper waves behind
	// workgroup barriers) the shape wins while every query gets a compute unit of its own (1M x 128: 404 against 737 us for
	// 256 queries, profiles/r03t_engine_shapes_by_batch_1m128_touch_threshold.txt).
	host::SearchShapePolicy shape_policy() const {
		host::SearchShapePolicy p;
		p.solo_mode = search_solo, p.solo_max_queries = solo_max_queries, p.solo_max_bytes = solo_max_bytes;
		p.team = search_team, p.n_cus = n_cus;
		p.touch_rows = search_touch_rows, p.touch_lists = search_touch_lists, p.touch_max_queries = search_touch_max_queries;
		p.force_looping = force_looping;
		p.team_box_bytes = TEAM_BOX_BYTES;
		p.crew = search_crew, p.engine_walkers = search_walkers;
		return p;
	}
	bool use_solo(uint32_t n) const {
		return host::wants_solo(shape_policy(), n, M0, V, G);
	}
	// searches over tombstones / a predicate start with the register queue (VSS_SEARCH_REG_QUEUE=0: always the unbounded one)
	bool search_reg_queue = true;
	uint32_t reg_queue_max_limit = 64 * MAX_LIST_REGS; // (round 2: 256; VSS_SEARCH_REG_QUEUE_MAX for A/B)
	uint32_t n_cus = 256;
	bool force_looping = false; // VSS_FORCE_LOOPING=1: the looping (NCH = 0) kernels for every dimension (A/B of the unrolled ones)
	uint32_t exact_probe = 0; // VSS_EXACT_PROBE: timing diagnostics of the score tile (answers are wrong with it set)
	// score tile: 1 = round 2's (single LDS buffer), 2 / 3 = software-pipelined 128x128 / 128x256 (round 3), 4 = the 128x128 tile as
	// persistent workgroups staged through registers (round 5), 5 = the same with its operands brought in by LDS-DMA (the
	// default; dimensions that are no multiple of 32 run 4)  (VSS_EXACT_KERNEL)
	uint32_t exact_kernel = 5;
	uint32_t HASH_LDS_MAX_LOG2 = 13;
	uint32_t BUILD_HASH_LDS_MAX_LOG2 = 11;
	// a table of this size cannot overflow: every node fits below the 7/8 fill limit
	uint32_t hash_max_log2() const {
		return std::max<uint32_t>(10, log2u((count + staged + 2) * 8 / 7 + 64));
	}
	// sizing rule: host_logic.h (visited_set_log2, search_cells_per_limit; CPU-tested).  vss_set_search_visited_set (or
	// VSS_VISITED_PER_LIMIT / VSS_HASH_LDS_MAX_LOG2 / VSS_VISITED_COMPACT, read ONCE in vss_create) override the cells per
	// limit entry, the largest table LDS takes and the compact form for A/B measurements and tests.
	bool retry_in_place = true;          // overflowing LDS sets: the walker repeats the query over a table in HBM (SearchArgs::retry_hash)
	uint64_t visited_per_limit = 0;      // 0 = the rule's own
	uint32_t hash_lds_max_override = 0;  // 0 = HASH_LDS_MAX_LOG2
	bool visited_compact_on = true;
	uint32_t hash_log2_for(uint64_t limit, uint32_t bump, uint64_t per_limit = 64) const {
		if (visited_per_limit)
			per_limit = visited_per_limit;
		return host::visited_set_log2(limit, bump, M0, list_cap_max(), per_limit, hash_max_log2());
	}
	DevBuf<uint32_t> d_global_hash;
	uint32_t *global_hash_for(uint32_t hash_log2, uint64_t grid) {
		if (hash_log2 <= BUILD_HASH_LDS_MAX_LOG2)
			return nullptr;
		d_global_hash.ensure(grid << hash_log2, 0, stream);
		return d_global_hash.p;
	}

	void ensure_build_scratch(uint64_t batch, uint64_t max_lv) {
		// requests per node: <= M per level
		const uint64_t req_cap = batch * M * (max_lv + 1) + 64;
		d_req_list.ensure(req_cap, 0, stream), d_req_src.ensure(req_cap, 0, stream);
		d_req_rank.ensure(req_cap, 0, stream), d_req_d.ensure(req_cap, 0, stream);
		d_sorted_src.ensure(req_cap, 0, stream), d_sorted_d.ensure(req_cap, 0, stream);
		d_touched.ensure(req_cap, 0, stream);
		const uint64_t lists = capacity + d_list_owner.n;
		d_list_count.ensure(lists, 0, stream, 0);
		d_list_offset.ensure(lists, 0, stream, 0);
		d_counters.ensure(8, 0, stream, 0);
		d_work_stats.ensure(4, 0, stream, 0);
		if (!h_counters)
			HIP_TRY(hipHostMalloc((void **)&h_counters, 8 * sizeof(uint32_t), hipHostMallocDefault));
	}

	int build_finalize() {
		if (!staged && !n_pending)
			return VSS_OK;
		if (refuse_while_probing("vss_build_finalize") != VSS_OK)
			return VSS_ERROR;
		const bool reuse = !st_slot.empty(); // explicit row order: some rows take over tombstoned slots
		lists_may_repeat |= reuse;
		const uint64_t first = count, n = reuse ? st_slot.size() : staged;
		std::vector<uint8_t> lv_rows;
		if (reuse) {
			lv_rows.resize(n);
			for (uint64_t i = 0; i != n; ++i)
				lv_rows[i] = levels_h[st_slot[i]];
			d_row_slot.ensure(n, 0, stream), d_row_src.ensure(n, 0, stream);
			HIP_TRY(hipMemcpyAsync(d_row_slot.p, st_slot.data(), n * 4, hipMemcpyHostToDevice, stream));
			HIP_TRY(hipMemcpyAsync(d_row_src.p, st_src.data(), n * 4, hipMemcpyHostToDevice, stream));
		}
		const uint8_t *lv = reuse ? lv_rows.data() : levels_h.data() + first;
		auto slot_of = [&](uint64_t row) { return reuse ? (uint64_t)st_slot[row] : first + row; };
		uint64_t solo_row = ~0ull;
		for (uint64_t i = 0; reuse && i != n; ++i)
			if (st_src[i] != EMPTY_SLOT && st_slot[i] == entry)
				solo_row = i;
		std::vector<uint64_t> sizes = batch_schedule(first, max_level, lv, n, max_batch, growth_div, solo_row);
		uint64_t biggest = 0, top_lv = 0;
		for (uint64_t b : sizes)
			biggest = std::max(biggest, b);
		for (uint64_t i = 0; i != n; ++i)
			top_lv = std::max<uint64_t>(top_lv, lv[i]);
		ensure_build_scratch(biggest, top_lv);
		uint64_t appended = 0; // rows linked so far that extended the node count
		progress_linked.store(0, std::memory_order_relaxed);
		progress_total.store(n, std::memory_order_relaxed);
		// list ids of upper lists are offset by the capacity: the per-list scratch must cover them
		uint64_t done = 0;
		int rc = VSS_OK;
		ensure_events();
		bool pending_b = false;
		auto wall0 = std::chrono::steady_clock::now();
		for (uint64_t b : sizes) {
			const uint64_t slot0 = slot_of(done);
			if (slot0 == 0 && first == 0 && done == 0) { // the very first node just becomes the entry point (index.hpp:2749-2753)
				entry = 0;
				max_level = levels_h[0];
				done += b;
				count = first + (appended += b);
				progress_linked.store(done, std::memory_order_relaxed);
				continue;
			}
			uint32_t bump = 0;
			uint32_t level_hi = 0, batch_reused = 0;
			for (uint64_t j = 0; j != b; ++j) {
				level_hi = std::max<uint32_t>(level_hi, lv[done + j]);
				batch_reused += reuse && st_src[done + j] != EMPTY_SLOT;
			}
			BuildArgs reuse_args;
			if (batch_reused) { // update(): blank the tapes of the reused nodes before anybody searches (index.hpp:2837-2840)
				reuse_args.gv = view();
				reuse_args.levels = d_levels.p;
				reuse_args.max_level = max_level;
				reuse_args.list_cap_max = list_cap_max();
				reuse_args.row_slot = d_row_slot.p + done, reuse_args.row_src = d_row_src.p + done;
				reuse_args.parked_stride = (level_hi + 1) * list_cap_max();
				d_parked.ensure((uint64_t)b * reuse_args.parked_stride, 0, stream);
				reuse_args.parked = d_parked.p;
				hipLaunchKernelGGL(k_reuse_lists, dim3((uint32_t)b), dim3(64), 0, stream, reuse_args, 0);
			}
			HIP_TRY(hipMemsetAsync(d_counters.p, 0, 8 * sizeof(uint32_t), stream));
			d_node_status.ensure(b, 0, stream);
			if (h_node_status.size() < b)
				h_node_status.resize(b);
			std::vector<uint32_t> work;
			for (;;) {
				BuildArgs a;
				a.gv = view();
				a.first_slot = (uint32_t)slot0;
				a.n_nodes = (uint32_t)b;
				a.levels = d_levels.p;
				a.entry = entry;
				a.max_level = max_level;
				a.top_limit = top_limit();
				a.hash_log2 = hash_log2_for(top_limit(), bump);
				a.list_cap_max = list_cap_max();
				a.req_list = d_req_list.p, a.req_src = d_req_src.p, a.req_d = d_req_d.p;
				a.counters = d_counters.p;
				a.req_capacity = (uint32_t)d_req_list.n;
				const uint32_t grid = work.empty() ? (uint32_t)b : (uint32_t)work.size();
				a.global_hash = global_hash_for(a.hash_log2, grid);
				a.work = work.empty() ? nullptr : d_work_build.p;
				a.node_status = d_node_status.p;
				a.node_req_cap = (uint32_t)(M * (level_hi + 1));
				a.work_stats = d_work_stats.p;
				a.row_slot = reuse ? d_row_slot.p + done : nullptr;
				a.row_src = reuse ? d_row_src.p + done : nullptr;
				a.pending = reinterpret_cast<const float4 *>(d_pending.p);
				a.parked = batch_reused ? d_parked.p : nullptr;
				a.parked_stride = (level_hi + 1) * list_cap_max();
				const bool list_in_hbm = top_limit() > 64 * MAX_LIST_REGS; // ef_construction beyond the register lists
				a.cand_lds_cap = list_in_hbm ? 16 : a.top_limit;
				a.list_cap = list_in_hbm ? a.top_limit : 0;
				a.list_buf = nullptr;
				if (list_in_hbm) {
					d_build_list.ensure((uint64_t)grid * 2 * a.list_cap, 0, stream);
					a.list_buf = d_build_list.p;
				}
				const uint32_t lds = wave_lds_bytes(a.hash_log2, V, a.list_cap_max, a.cand_lds_cap, !a.global_hash) +
				                     align16(a.node_req_cap * 4) * 2;
				HIP_TRY(hipEventRecord(ev[0], stream));
				launch_by_metric<BuildArgs>(launch_phase_a<0>, launch_phase_a<1>, launch_phase_a<2>, a,
				                            launch_cfg(grid, lds, top_limit()));
				HIP_TRY(hipEventRecord(ev[1], stream));
				HIP_TRY(hipMemcpyAsync(h_counters, d_counters.p, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
				HIP_TRY(hipStreamSynchronize(stream));
				{
					float ms = 0;
					HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
					timing[1] += ms;
					if (pending_b) { // the previous batch's link kernels finished before this phase A started
						HIP_TRY(hipEventElapsedTime(&ms, ev[2], ev[3]));
						timing[2] += ms;
						pending_b = false;
					}
					timing[4] += 1;
				}
				if (!h_counters[3])
					break;
				// some nodes overflowed their visited set: re-run just those with a larger table
				if (a.hash_log2 >= hash_max_log2()) {
					rc = fail("visited-set overflow during build");
					break;
				}
				HIP_TRY(hipMemcpy(h_node_status.data(), d_node_status.p, b * 4, hipMemcpyDeviceToHost));
				std::vector<uint32_t> failed;
				if (work.empty()) {
					for (uint32_t j = 0; j != b; ++j)
						if (h_node_status[j])
							failed.push_back(j);
				} else {
					for (uint32_t j : work)
						if (h_node_status[j])
							failed.push_back(j);
				}
				work.swap(failed);
				timing[5] += work.size();
				d_work_build.ensure(b, 0, stream);
				HIP_TRY(hipMemcpyAsync(d_work_build.p, work.data(), work.size() * 4, hipMemcpyHostToDevice, stream));
				HIP_TRY(hipMemsetAsync(d_counters.p + 3, 0, sizeof(uint32_t), stream));
				bump += 2;
			}
			if (rc != VSS_OK)
				break;
			if (batch_reused)
				hipLaunchKernelGGL(k_reuse_lists, dim3((uint32_t)b), dim3(64), 0, stream, reuse_args, 1);
			const uint32_t n_req = h_counters[0];
			if (n_req) {
				LinkArgs l;
				l.gv = view();
				l.req_list = d_req_list.p, l.req_src = d_req_src.p, l.req_d = d_req_d.p, l.req_rank = d_req_rank.p;
				l.counters = d_counters.p;
				l.list_count = d_list_count.p, l.list_offset = d_list_offset.p, l.touched = d_touched.p;
				l.sorted_src = d_sorted_src.p, l.sorted_d = d_sorted_d.p;
				l.list_owner = d_list_owner.p, l.upper_off = d_upper_off.p;
				l.hash_log2 = 4; // phase B needs no visited set
				l.list_cap_max = list_cap_max();
				l.work_stats = d_work_stats.p;
				const uint32_t tb = 256, gb = std::min<uint32_t>((n_req + tb - 1) / tb, 2048);
				HIP_TRY(hipEventRecord(ev[2], stream));
				hipLaunchKernelGGL(k_link_count, dim3(gb), dim3(tb), 0, stream, l);
				hipLaunchKernelGGL(k_link_alloc, dim3(gb), dim3(tb), 0, stream, l);
				hipLaunchKernelGGL(k_link_scatter, dim3(gb), dim3(tb), 0, stream, l);
				const uint32_t lds = wave_lds_bytes(l.hash_log2, V, l.list_cap_max, l.list_cap_max + 1);
				launch_by_metric<LinkArgs>(launch_phase_b<0>, launch_phase_b<1>, launch_phase_b<2>, l,
				                           launch_cfg(std::min<uint32_t>(n_req, 65536), lds, 64));
				HIP_TRY(hipEventRecord(ev[3], stream));
				pending_b = true;
			}
			// update() epilogue: the new key, then the new vector (index.hpp:2850, index_dense.hpp:1777-1781) — stream
			// ordered after the link kernels, which still measured against the old vector
			for (uint64_t j = 0; batch_reused && j != b; ++j) {
				const uint32_t src = st_src[done + j], slot = st_slot[done + j];
				if (src == EMPTY_SLOT)
					continue;
				const uint64_t stride = (uint64_t)V * 4;
				HIP_TRY(hipMemcpyAsync(d_vectors.p + slot * stride, d_pending.p + src * stride, stride * 4,
				                       hipMemcpyDeviceToDevice, stream));
				HIP_TRY(hipMemcpyAsync(d_keys.p + slot, &pending_keys[src], 8, hipMemcpyHostToDevice, stream));
			}
			// a node above the current top level is always a singleton batch: it becomes the entry (index.hpp:2769-2772)
			for (uint64_t j = 0; j != b; ++j) {
				if ((int)lv[done + j] > max_level) {
					max_level = lv[done + j];
					entry = (uint32_t)slot_of(done + j);
				}
			}
			for (uint64_t j = 0; j != b; ++j)
				appended += !reuse || st_src[done + j] == EMPTY_SLOT;
			tombstones -= batch_reused; // their keys are live on the device from here on (stream order)
			done += b;
			count = first + appended;
			progress_linked.store(done, std::memory_order_relaxed);
		}
		HIP_TRY(hipStreamSynchronize(stream));
		if (pending_b) {
			float ms = 0;
			HIP_TRY(hipEventElapsedTime(&ms, ev[2], ev[3]));
			timing[2] += ms;
		}
		timing[3] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
		// rows of batches that did not run (error): plain appends stay staged; rows that were to take over a tombstoned slot
		// hand it back (the slot stays a tombstone: key, ring and rowid map as before the call)
		for (uint64_t i = done; reuse && i < n; ++i) {
			if (st_src[i] == EMPTY_SLOT)
				continue;
			const uint32_t slot = st_slot[i];
			if (keymap.ready)
				keymap.erase(keys_h[slot]);
			keys_h[slot] = VSS_FREE_KEY;
			if (free_slots.reserve(free_slots.size() + 1))
				free_slots.push(slot);
		}
		staged = first + staged - count;
		st_slot.clear(), st_src.clear(), pending_keys.clear();
		n_pending = 0;
		mutations++;
		// optional: finish a bulk build with the reference's compaction order (vss_set_build_reorder) — nodes of one cluster
		// contiguous in HBM; only when nothing is tombstoned (a build never prunes behind the caller's back)
		if (rc == VSS_OK && reorder_after_build && !tombstones && !staged && first == 0)
			rc = compact(true);
		return rc;
	}

	// ------------------------------------------------------------------ search
	SearchCtx &context(int slot) {
		SearchCtx &c = ctx[slot];
		if (!c.stream) {
			if (slot == 0) {
				c.stream = stream;
			} else {
				HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
				c.own_stream = true;
			}
			HIP_TRY(hipEventCreate(&c.ev0));
			HIP_TRY(hipEventCreate(&c.ev1));
			// (coherent whatever HIP_HOST_COHERENT says: the host spins on these words while the kernel that writes them still runs)
			HIP_TRY(hipHostMalloc((void **)&c.h_queue, 16, hipHostMallocCoherent));
		}
		if (slot == 0)
			c.stream = stream; // follows vss_set_stream
		return c;
	}

	// Launch the search engine over `n` queries (all of the batch, or the entries of args.work in a retry pass).
	void launch_search_kernel(SearchCtx &c, uint32_t n) {
		SearchArgs &a = c.args;
		a.n_queries = n;
		const host::SearchShapePolicy policy = shape_policy();
		const bool solo = host::wants_solo(policy, n, M0, V, G);
		// One walker per compute unit — the solo shape, or the workgroup engine on a launch of at most one query per compute
		// unit — has the unit's LDS to itself: a visited set four times roomier (at most 64 KiB) keeps the probe sequences of
		// a chunk of 64 ids short — the gather phase is dominated by them
		const bool roomy = host::roomy_visited_set(policy, solo, n); // (host_logic.h: the rule choose_search_shape reports)
		a.hash_log2 = hash_log2_for(c.limit, c.bump, host::search_cells_per_limit(c.limit));
		if (roomy && a.hash_log2 <= HASH_LDS_MAX_LOG2)
			a.hash_log2 = std::min<uint32_t>({a.hash_log2 + 2, 14u, std::max(a.hash_log2, hash_max_log2())});
		// a retry after a visited-set overflow must get a LARGER table than the one that overflowed, whatever shape and
		// enlargement the pass before had (c.min_hash_log2 = that table's size + 1; nothing exceeds "every node fits")
		a.hash_log2 = std::max(a.hash_log2, std::min(c.min_hash_log2, hash_max_log2()));
		// LDS or HBM: tables up to 32 KiB stay in LDS (four walkers per workgroup).  (Measured and not kept: 64-KiB tables in LDS
		// with two walkers — 0.77 against 0.60 of the HBM peak at 12.5M x 1536 / ef 192, but 0.45 against 0.56 at 768 dims and
		// 0.51 against 0.63 on the configs[4] shard at ef 480: two walkers do not feed a compute unit once expansions are thin.)
		const uint32_t hash_lds_max = hash_lds_max_override ? hash_lds_max_override : HASH_LDS_MAX_LOG2;
		bool hash_in_lds = a.hash_log2 <= (roomy ? 14u : hash_lds_max);
		// Limits of 257-512 — the 8-register list's instantiation — whose 32-bit table would go to HBM take the COMPACT exact
		// form instead (16-bit cells: tag + displacement, wave_primitives.h) over the largest table LDS admits: twice the cells
		// in the same bytes, every probe an LDS round trip instead of an L2 / memory one (2M x 1536 at ef 480 / 320: 0.52 -> 0.60
		// / 0.59 -> 0.68 of the HBM peak, answers and work counters identical: profiles/r04_compact_visited_set_*.txt).  Slots
		// must fit 24 bits; first pass only — a query that overflows it (too many visits, or a displacement beyond its bits)
		// is re-run with the plain table like any other overflow.  vss_set_search_visited_set(index, 0, ..) turns it off for A/B.
		// (the rule: host_logic.h, CPU-tested)
		a.visited_compact = 0;
		{
			const uint32_t lds_table_log2 = roomy ? 14u : hash_lds_max;
			const uint32_t cells_log2 = visited_compact_on ? host::compact_visited_cells_log2(hash_in_lds, solo, !c.list_cap, c.limit, count,
			                                                                                  !c.bump && !c.min_hash_log2, lds_table_log2)
			                                                : 0u;
			if (cells_log2) {
				a.hash_log2 = lds_table_log2;
				a.visited_compact = cells_log2;
				hash_in_lds = true;
			}
		}
		// walkers per workgroup: as many as the batch needs to cover every compute unit once, as many as LDS admits
		uint32_t waves = std::max<uint32_t>(2, std::min<uint32_t>(search_waves, 16));
		// the accept phase of an expansion in the shadow of the successor's row loads (level_search_pipelined): plain searches
		// with a register list over neighbour lists of at most 64 cells.  Limits of 257-512 (the 8-register list) are pipelined
		// in 12-wave workgroups only (170 registers per lane: k_search<.., WIDE_LIST_THREADS>, round 5) — four walkers and eight
		// scoring waves; vss_set_search_wide_lists(0) keeps them in 16 waves and the plain order (round 4; A/B)
		const bool can_pipeline = search_pipelined && !solo && !a.tomb && !c.list_cap && list_cap_max() <= 64;
		const bool wide_list = can_pipeline && search_wide_lists && c.limit > 64u * PIPELINED_MAX_REGS;
		if (wide_list)
			waves = std::min<uint32_t>(waves, WIDE_LIST_THREADS / 64);
		// (rounds 2-5 staged the batched list merge here — `limit` cells per walker; the blocked list of round 6 merges nothing.
		//  What is left is four words per walker of the 8-register list's kernels: where its visited set moves when it outgrows
		//  LDS, WaveLds::spill_box)
		a.stage_cap = (!solo && !c.list_cap && c.limit > 64u * PIPELINED_MAX_REGS) ? 4u : 0u;
		const uint32_t slot_bytes = engine_slot_bytes(a.hash_log2, V, a.list_cap_max, hash_in_lds, a.stage_cap);
		uint32_t s_max = std::min<uint32_t>({search_walkers ? ENGINE_MAX_WALKERS : search_walkers_cap, waves - 1,
		                                     (160u * 1024 - ENGINE_HEADER_BYTES) / slot_bytes});
		uint32_t S = search_walkers ? search_walkers : (n + n_cus - 1) / n_cus;
		S = std::max<uint32_t>(1, std::min(S, s_max));
		// workgroups per compute unit: one 16-wave workgroup — or, with fewer waves per workgroup (vss_set_search_params), as
		// many as the waves and the LDS admit, so that a workgroup of the NEXT launch can move in beside one that still drains
		// (VSS_SEARCH_WGS_PER_CU caps it, default 1.  Measured: two 8-wave workgroups of 2 walkers + 6 scoring waves per compute
		// unit run every regime at the rate of one 16-wave workgroup of 4 + 12 — profiles/r04_two_workgroups_per_compute_unit_no_gain.txt)
		uint32_t wgs_per_cu = std::max<uint32_t>(1, std::min<uint32_t>(16u / waves, (160u * 1024) /
		                                             std::max<uint32_t>(1, engine_lds_bytes(S, a.hash_log2, V, a.list_cap_max, hash_in_lds, a.stage_cap))));
		wgs_per_cu = std::min(wgs_per_cu, search_wgs_per_cu);
		uint32_t grid = std::min<uint32_t>(n_cus * wgs_per_cu, (n + S - 1) / S);
		const uint32_t solo_lds = wave_lds_bytes(a.hash_log2, V, a.list_cap_max, a.stage_cap, hash_in_lds);
		if (solo) { // one single-wave workgroup per query in flight, as many per compute unit as LDS admits (at most 8)
			S = 1;
			grid = std::min<uint32_t>(n, n_cus * std::max<uint32_t>(1, std::min<uint32_t>(8, 160u * 1024 / solo_lds)));
		}
		// scratch in HBM scales with the resident walkers: bound it (retry passes with very large tables run fewer at a time)
		// a query that outgrows its LDS-resident set is repeated by its walker, in the same launch, over a table in HBM (2^17
		// cells = 512 KiB per walker, or what holds the whole index if that is less) — where overflows are a per-cent matter:
		// limits of 257-512, the compact form (the only instantiations that carry the code: k_search<.., 8, ..>)
		uint32_t retry_want_log2 = 0;
		if (retry_in_place && !solo && hash_in_lds && !c.list_cap && c.limit > 64u * PIPELINED_MAX_REGS) { // (the 8-register list's kernels)
			const uint32_t have_log2 = a.visited_compact ? compact_visited::cells_log2_of(a.visited_compact) : a.hash_log2;
			const uint32_t want_log2 = std::min<uint32_t>(17u, hash_max_log2());
			if (want_log2 > have_log2)
				retry_want_log2 = want_log2;
		}
		// (the retry tables count: ADVICE r05 — 512 MiB per context that runs such limits went unbudgeted)
		const uint64_t per_walker = (hash_in_lds ? 0 : (4ull << a.hash_log2)) + 8ull * c.list_cap + (a.tomb == 2 ? 8ull * c.cand_cap : 0) +
		                            (retry_want_log2 ? (4ull << retry_want_log2) : 0);
		const uint64_t budget = 16ull << 30;
		while (per_walker * grid * S > budget && (grid > 1 || S > 1)) {
			if (S > 1)
				S--;
			else
				grid = (grid + 1) / 2;
		}
		a.walkers = S;
		// one expansion of look-ahead while scoring waves are idle (lists of at most 64 cells: one cell per lane)
		a.spec_active = (!a.tomb && !solo && list_cap_max() <= 64) ? search_spec_active : 0;
		// latency-bound launches of the solo shape (at most one query per compute unit): pull the rows the cached lists name
		// (RowTouch: by the team's helpers, or by the lone wave for rows of at most four 128-byte lines) and the lists of the
		// rows being accepted (ListTouch) into L2 ahead of time.  Costs bandwidth, so never for launches that could be bound by it.
		const host::SearchShape shape = host::choose_search_shape(policy, n, M0, V, G, solo_lds);
		a.touch_lines = shape.touch_lines;
		// the last walker of a workgroup runs its scoring waves as a crew (two barriers per expansion instead of the mailbox
		// exchange): from the start when S = 1, in the drain of a larger launch otherwise
		a.crew = shape.crew ? (CREW_ON | ((search_touch_lists && list_cap_max() <= 64) ? CREW_TOUCH : 0u) | search_crew_tune) : 0u;
		// the accept phase of an expansion in the shadow of the successor's row loads (level_search_pipelined): plain searches
		// with a register list over neighbour lists of at most 64 cells
		a.pipelined = (can_pipeline && (wide_list || c.limit <= 64u * PIPELINED_MAX_REGS)) ? 1u : 0u;
		a.global_hash = nullptr;
		if (!hash_in_lds) {
			c.d_global_hash.ensure(((uint64_t)grid * S) << a.hash_log2, 0, c.stream);
			a.global_hash = c.d_global_hash.p;
		}
		// the retry tables: no memory for them is no reason to fail a search that would have succeeded without — such queries
		// then go back to the host's re-run launch, as before round 5
		a.retry_hash = nullptr, a.retry_log2 = 0;
		if (retry_want_log2 && c.d_retry_hash.try_ensure(((uint64_t)grid * S) << retry_want_log2))
			a.retry_hash = c.d_retry_hash.p, a.retry_log2 = retry_want_log2;
		a.list_cap = (uint32_t)c.list_cap;
		a.list_buf = nullptr;
		if (c.list_cap) {
			c.d_list_buf.ensure((uint64_t)grid * S * 2 * c.list_cap, 0, c.stream);
			a.list_buf = c.d_list_buf.p;
		}
		a.cand_cap = c.cand_cap;
		a.cand_buf = nullptr;
		if (a.tomb == 2) {
			c.d_cand_buf.ensure((uint64_t)grid * S * 2 * c.cand_cap, 0, c.stream);
			a.cand_buf = c.d_cand_buf.p;
		}
		if (!c.d_queue.p) { // first launch on this context: both query counters start at zero
			c.d_queue.ensure(4 + 64, 0, c.stream);
			HIP_TRY(hipMemsetAsync(c.d_queue.p, 0, (4 + 64) * sizeof(uint32_t), c.stream));
			c.queue_sel = 0;
		}
		a.queue = c.d_queue.p;
		a.queue_sel = c.queue_sel;
		c.queue_sel ^= 2u;
		c.flag_wait = c.direct_io && probe_flag_wait;
		if (c.unsynced && !c.flag_wait) { // the previous launch was waited for by flag: let it retire before its words are reset
			HIP_TRY(hipStreamSynchronize(c.stream));
			c.unsynced = false;
		}
		a.engine_error = c.h_queue + 1;
		if (c.flag_wait) {
			// a chain of flag-waited launches never resets a pinned word (the previous kernel may still be retiring): the answer
			// count only grows, the error word is sticky, and nobody gates on such a launch
			if (!c.unsynced)
				c.h_queue[1] = 0, c.h_queue[3] = 0;
			a.drain_flag = nullptr;
			a.done_count = c.h_queue + 3;
		} else {
			c.h_queue[1] = 0, c.h_queue[2] = 0; // (this context has no launch in flight: nobody is writing them)
			a.drain_flag = c.h_queue + 2;
			a.done_count = nullptr;
		}
		LaunchCfg cfg = launch_cfg(grid, solo ? solo_lds : engine_lds_bytes(S, a.hash_log2, V, a.list_cap_max, hash_in_lds, a.stage_cap),
		                           c.limit);
		cfg.stream = c.stream;
		// a team (helper waves on the compute unit's other SIMDs score a share of every expansion's rows) while every query
		// of the launch still gets a compute unit of its own; its job box takes the first bytes of the workgroup's LDS
		const bool team = shape.team;
		if (team)
			cfg.lds += TEAM_BOX_BYTES;
		cfg.threads = solo ? (team ? 64 * VSS_TEAM_WAVES : 64) : 64 * waves;
		if (c.flag_wait)
			c.flag_t0 = std::chrono::steady_clock::now();
		else
			HIP_TRY(hipEventRecord(c.ev0, c.stream));
		if (solo)
			launch_by_metric<SearchArgs>(launch_search_solo<0>, launch_search_solo<1>, launch_search_solo<2>, a, cfg);
		else
			launch_by_metric<SearchArgs>(launch_search<0>, launch_search<1>, launch_search<2>, a, cfg);
		if (c.flag_wait) // committed only now: a launch that failed above must not leave a target no kernel will ever reach
			c.flag_target = (c.unsynced ? c.flag_target : 0u) + n;
		else
			HIP_TRY(hipEventRecord(c.ev1, c.stream));
		if (!c.direct_io) {
			HIP_TRY(hipMemcpyAsync(c.h_status, c.d_status.p, c.nq * 4, hipMemcpyDeviceToHost, c.stream));
			HIP_TRY(hipMemcpyAsync(c.h_stats, c.d_stats.p, c.nq * 8, hipMemcpyDeviceToHost, c.stream));
		}
	}

	// Pipelined launches (explicit contexts): a launch of the engine occupies every compute unit, so a second one issued
	// right away would only sit in its queue — and its clock (events, profilers) would run while it waits.  It is issued
	// when the previous one has handed out its last query, i.e. when compute units start to fall idle: the same overlap of
	// one launch's tail with the next one's body, and launch durations that mean execution.  Never waits longer than
	// the previous launch runs (its completion event ends the wait as well).
	bool search_gating = true;
	void wait_for_drain_of_previous_launch(int slot);
	// the pinned small-batch probe waits on a completion word in host memory (1) or on its stream, with events (0)
	bool probe_flag_wait = true;
	// solo shape: touch the rows of the cached neighbour lists one expansion ahead (launches of at most this many queries)
	bool search_touch_rows = true, search_touch_lists = true;
	uint32_t search_touch_max_queries = 256; // (measured up to one query per compute unit: 404 against 456 us for 256 queries)
	// solo shape with helper waves (teams), VSS_SEARCH_TEAM=0 for A/B
	bool search_team = true;
	// workgroup engine: crew mode for the last walker of a workgroup (vss_set_search_crew, VSS_SEARCH_CREW=0 for A/B)
	bool search_crew = true;
	// crew refinements (CREW_SPARE_SIMD | CREW_NO_REQUESTS bits; VSS_SEARCH_CREW_TUNE for A/B).  Both measured inside the
	// noise at 3M x 768 (profiles/r04d_*): off.
	uint32_t search_crew_tune = 0;
	// workgroup engine: software-pipelined level search (vss_set_search_pipelined, VSS_SEARCH_PIPELINED=0 for A/B)
	bool search_pipelined = true;
	// limits of 257-512 pipelined in 12-wave workgroups (vss_set_search_wide_lists, VSS_SEARCH_WIDE_LISTS=0 for A/B)
	bool search_wide_lists = true;

	// enqueue one batched probe on a context (asynchronous); search_end() completes it
	int search_begin(int slot, const float *d_queries, uint32_t q_stride, uint64_t nq, uint64_t k, uint64_t ef,
	                 int64_t *d_keys_out, float *d_dist_out, uint32_t *d_count_out,
	                 const uint64_t *d_filter = nullptr, uint64_t filter_bits = 0, bool direct_io = false) {
		return search_begin_multi(slot, 1, &d_queries, q_stride, nq, k, ef, &d_keys_out, &d_dist_out, &d_count_out, d_filter,
		                          filter_bits, direct_io);
	}

	// The same for `n_batches` probe batches of `per_batch` queries each, answered by ONE launch of the search engine:
	// its walkers take queries from all of them until none is left, so compute units that drew short queries in one batch
	// go on with the next instead of idling until the slowest query of the batch is done.
	int search_begin_multi(int slot, uint64_t n_batches, const float *const *d_queries, uint32_t q_stride, uint64_t per_batch,
	                       uint64_t k, uint64_t ef, int64_t *const *d_keys_out, float *const *d_dist_out,
	                       uint32_t *const *d_count_out, const uint64_t *d_filter = nullptr, uint64_t filter_bits = 0,
	                       bool direct_io = false, bool gate = false) {
		if (slot < 0 || slot >= MAX_CTX)
			return fail("search context %d out of range (0..%d)", slot, MAX_CTX - 1);
		if (n_batches < 1 || n_batches > MAX_COALESCED)
			return fail("1 to %d batches per launch", MAX_COALESCED);
		SearchCtx &c = context(slot);
		if (c.pending)
			return fail("search context %d already has a batch in flight", slot);
		if (gate)
			wait_for_drain_of_previous_launch(slot);
		if (!ef)
			ef = efs ? efs : 64;
		const uint64_t limit = std::max(ef, k);
		if (k > 0x7FFFFFFFull || ef > 0x7FFFFFFFull)
			return fail("k / ef_search above 2^31-1");
		const uint64_t nq = n_batches * per_batch;
		if (nq > 0x7FFFFFFFull)
			return fail("more than 2^31-1 queries in one launch");
		c.stats[0] = c.stats[1] = c.stats[3] = 0;
		c.stats[2] = nq;
		c.kernel_ms = 0;
		c.nq = nq;
		if (!nq || !k)
			return VSS_OK;
		if (!count) { // empty index: no results (index.hpp:2895-2896)
			for (uint64_t b = 0; b != n_batches; ++b) {
				HIP_TRY(hipMemsetAsync(d_keys_out[b], 0xFF, per_batch * k * 8, c.stream));
				if (d_dist_out[b])
					HIP_TRY(hipMemsetAsync(d_dist_out[b], 0x7F, per_batch * k * 4, c.stream));
				HIP_TRY(hipMemsetAsync(d_count_out[b], 0, per_batch * 4, c.stream));
			}
			c.nq = 0;
			c.pending = true;
			return VSS_OK;
		}
		// the context's per-query words: sized ONCE for the largest launch the boundary admits at this batch size (a context that
		// has carried several batches will carry 32 — 12 bytes per query), not re-grown — device and pinned reallocations, each a
		// device-wide synchronisation — whenever a launch carries more batches than its predecessors (round 6: the driver's
		// 5 warm-up steps followed by 20 timed ones put exactly that inside the timed region: 2 % of it)
		const uint64_t cap_q = n_batches > 1 ? std::max<uint64_t>(nq, (uint64_t)MAX_COALESCED * per_batch) : nq;
		c.d_stats.ensure(cap_q * 2, 0, c.stream);
		c.d_status.ensure(cap_q, 0, c.stream);
		if (c.h_cap < nq) {
			if (c.h_status)
				(void)hipHostFree(c.h_status), (void)hipHostFree(c.h_stats);
			HIP_TRY(hipHostMalloc((void **)&c.h_status, cap_q * 4, hipHostMallocCoherent));
			HIP_TRY(hipHostMalloc((void **)&c.h_stats, cap_q * 8, hipHostMallocCoherent));
			c.h_cap = cap_q;
		}
		SearchArgs &a = c.args;
		a.gv = view();
		a.gv.filter = reinterpret_cast<const unsigned long long *>(d_filter);
		a.gv.filter_bits = filter_bits;
		for (uint64_t b = 0; b != MAX_COALESCED; ++b) {
			const bool used = b < n_batches;
			a.queries[b] = used ? d_queries[b] : nullptr;
			a.out_keys[b] = used ? d_keys_out[b] : nullptr;
			a.out_d[b] = used ? d_dist_out[b] : nullptr;
			a.out_count[b] = used ? d_count_out[b] : nullptr;
		}
		a.batch_size = (uint32_t)per_batch;
		a.q_stride = q_stride;
		a.n_queries = (uint32_t)nq;
		a.k = (uint32_t)k;
		a.ef = (uint32_t)ef;
		a.entry = entry;
		a.max_level = max_level;
		// rejected rows (tombstones, predicate) are traversed but not returned: the pending candidates wait in a register
		// queue while the limit allows (RegQueue), in the unbounded queue in HBM otherwise or once a query outgrew the former
		a.tomb = (tombstones || d_filter) ? (limit <= reg_queue_max_limit && search_reg_queue ? 1u : 2u) : 0u;
		a.list_cap_max = list_cap_max();
		a.work = nullptr;
		c.direct_io = direct_io;
		a.out_stats = direct_io ? c.h_stats : c.d_stats.p;
		a.status = direct_io ? c.h_status : c.d_status.p;
		a.phase_ticks = nullptr;
#ifdef VSS_PHASE_TIMERS
		c.d_phase.ensure(nq * VSS_PHASE_STRIDE, 0, c.stream);
		a.phase_ticks = c.d_phase.p;
#endif
		c.limit = limit;
		c.bump = 0;
		c.min_hash_log2 = 0;
		// beyond the register lists (ef_search or k above 512) the candidate list lives in HBM; no list outgrows the index
		c.list_cap = limit > 64 * MAX_LIST_REGS ? std::min<uint64_t>(limit, count) : 0;
		// accepted-but-unexpanded candidates of a search over tombstones / a predicate (the reference's unbounded `next` heap);
		// a query that outgrows the queue is re-run with a larger one
		c.cand_cap = (uint32_t)std::min<uint64_t>(count + 1, std::max<uint64_t>(1024, 4 * limit));
		if (direct_io)
			std::memset(c.h_status, 0xFF, nq * 4); // "not processed" until the engine says otherwise
		else
			HIP_TRY(hipMemsetAsync(c.d_status.p, 0xFF, nq * 4, c.stream));
		launch_search_kernel(c, (uint32_t)nq);
		c.pending = true;
		return VSS_OK;
	}

	int search_end(int slot, bool keep_query_stats) {
		if (slot < 0 || slot >= MAX_CTX)
			return fail("search context %d out of range", slot);
		SearchCtx &c = context(slot);
		if (!c.pending)
			return VSS_OK;
		c.pending = false;
		std::vector<uint32_t> work;
		int rounds = 0;
		for (;;) {
			if (c.flag_wait && c.nq) {
				// The one-query probe: ids, distances, counts and status are in pinned host memory, each query's published by a
				// system-scope release on h_queue[3].  Waiting on that word instead of on the stream saves the wake-up of
				// hipStreamSynchronize and the two event packets around the kernel; the stream is synchronised lazily, before
				// the context's next launch.  `search_kernel_ms` of such a call is the host clock from launch to flag.
				volatile uint32_t *done = c.h_queue + 3;
				bool seen = false;
				for (uint64_t spins = 0;; ++spins) {
					if ((int32_t)(*done - c.flag_target) >= 0) { // (wrap-safe: the count only grows)
						seen = true;
						break;
					}
					if ((spins & 1023) == 1023) {
						const auto waited = std::chrono::steady_clock::now() - c.flag_t0;
						if (waited > std::chrono::milliseconds(2)) { // far beyond any probe: let the stream decide
							if (hipStreamQuery(c.stream) != hipErrorNotReady)
								break; // finished without the flag (a failed launch), or done meanwhile: the stream decides
							std::this_thread::yield();
						}
					} else {
#if defined(__x86_64__) || defined(__i386__)
						__builtin_ia32_pause();
#else
						std::this_thread::yield();
#endif
					}
				}
				(void)hipGetLastError();
				std::atomic_thread_fence(std::memory_order_acquire);
				c.kernel_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c.flag_t0).count();
				if (seen) {
					c.unsynced = true;
				} else {
					HIP_TRY(hipStreamSynchronize(c.stream));
					c.unsynced = false; // (the next flag launch starts from zero again)
				}
			} else {
				HIP_TRY(hipStreamSynchronize(c.stream));
				c.unsynced = false;
			}
			if (!c.nq)
				break;
			if (!c.flag_wait) {
				float ms = 0;
				HIP_TRY(hipEventElapsedTime(&ms, c.ev0, c.ev1));
				c.kernel_ms += ms;
			}
			if (c.h_queue[1])
				return fail("search engine: a walking wave gave up waiting for its scoring waves (internal error)");
#ifdef VSS_PARANOID
			{
				const uint32_t *note = h_debug;
				if (note[0])
					return fail("paranoid: %u bad chunks; first: c %u n %u id %u block %u wave %u ticket {n %u next %u} cnt %u last id %u",
					            note[0], note[1], note[2], note[3], note[4], note[5], note[6], note[7], note[8], note[9]);
			}
#endif
			work.clear();
			bool visited_full = false, queue_full = false;
			for (uint64_t i = 0; i != c.nq; ++i) {
				if (!c.h_status[i])
					continue;
				if (c.h_status[i] == LEVEL_OK_RETRIED) { // answered — by its walker's second attempt, over the table in HBM
					c.h_status[i] = LEVEL_OK;
					// counted once, among the re-run queries: in the first round only.  The device's status array is copied
					// back whole after every launch, so the word of a query that is NOT part of a later re-run round shows up
					// again there; a re-run query that its walker repeated once more is in `work` already.
					if (rounds == 0)
						c.stats[3] += 1;
					continue;
				}
				if (c.h_status[i] > LEVEL_QUEUE_OVERFLOW)
					return fail("search engine: query %llu was not processed (status %u; internal error)", (unsigned long long)i,
					            c.h_status[i]);
				work.push_back((uint32_t)i);
				visited_full |= c.h_status[i] == LEVEL_VISITED_OVERFLOW;
				queue_full |= c.h_status[i] == LEVEL_QUEUE_OVERFLOW;
			}
			if (work.empty())
				break;
			// some queries outgrew their scratch: re-run just those with more of it.  Both sizes are bounded by the number
			// of nodes (a visited set or a queue that holds every node cannot overflow), so this always terminates.
			if ((visited_full && c.args.hash_log2 >= hash_max_log2()) ||
			    (queue_full && c.args.tomb == 2 && c.cand_cap >= count + 1))
				return fail("search scratch overflow although sized for the whole index");
			if (++rounds > 64)
				return fail("search engine: scratch retries do not converge (internal error)");
			if (visited_full) {
				c.bump += 2;
				c.min_hash_log2 = std::max(c.args.hash_log2, c.args.retry_log2) + 1; // larger than anything these queries have had
			}
			if (queue_full) {
				if (c.args.tomb == 1)
					c.args.tomb = 2; // outgrew the register queue: the unbounded one
				else
					c.cand_cap = (uint32_t)std::min<uint64_t>(count + 1, (uint64_t)c.cand_cap * 8);
			}
			c.stats[3] += work.size();
			c.d_work.ensure(c.nq, 0, c.stream);
			HIP_TRY(hipMemcpyAsync(c.d_work.p, work.data(), work.size() * 4, hipMemcpyHostToDevice, c.stream));
			HIP_TRY(hipStreamSynchronize(c.stream)); // `work` is reused
			c.args.work = c.d_work.p;
			launch_search_kernel(c, (uint32_t)work.size());
		}
		for (uint64_t i = 0; i != c.nq; ++i) {
			c.stats[0] += c.h_stats[2 * i];
			c.stats[1] += c.h_stats[2 * i + 1];
		}
		{
			std::lock_guard<std::mutex> lk(stats_mu);
			std::memcpy(last_stats, c.stats, sizeof last_stats);
			timing[0] = c.kernel_ms;
			if (keep_query_stats)
				last_query_stats.assign(c.h_stats, c.h_stats + 2 * c.nq);
		}
		return VSS_OK;
	}

	// blocking search on context `slot` (0 = the index stream, shared by the blocking device-pointer calls)
	int search_launch(int slot, const float *d_queries, uint32_t q_stride, uint64_t nq, uint64_t k, uint64_t ef,
	                  int64_t *d_keys_out, float *d_dist_out, uint32_t *d_count_out, bool keep_query_stats,
	                  const uint64_t *d_filter = nullptr, uint64_t filter_bits = 0, bool direct_io = false) {
		std::unique_lock<std::mutex> lk0(ctx0_mu, std::defer_lock);
		if (slot == 0)
			lk0.lock();
		int rc = search_begin(slot, d_queries, q_stride, nq, k, ef, d_keys_out, d_dist_out, d_count_out, d_filter,
		                      filter_bits, direct_io);
		if (rc != VSS_OK)
			return rc;
		return search_end(slot, keep_query_stats);
	}

	struct Lease { // a pooled context for the duration of one host-pointer call
		vss_index *ix;
		int slot;
		explicit Lease(vss_index *i) : ix(i), slot(i->lease_context()) {
		}
		~Lease() {
			ix->ctx[slot].pending = false;
			ix->return_context(slot);
		}
	};

	static constexpr uint64_t DIRECT_IO_MAX_QUERIES = 256;
	int search_host(const float *queries, uint64_t nq, uint64_t k, uint64_t ef, int64_t *out_keys, float *out_d,
	                uint32_t *out_counts, bool exact, const uint64_t *filter = nullptr, uint64_t filter_bits = 0) {
		if (!nq || !k)
			return VSS_OK;
		Lease lease(this);
		SearchCtx &c = context(lease.slot);
		// Small batches — the one-query probe of HNSW_INDEX_SCAN (hnsw_index.cpp:315-356) and the chunks of HNSW_INDEX_JOIN
		// (hnsw_optimize_join.cpp:111-168; 204 queries at 768 dims): the kernel reads the queries from, and writes ids /
		// distances / counts / status straight into, one pinned host block.  No staging copies; the only host-device
		// interaction is the launch and one synchronisation.  (Up to DIRECT_IO_MAX_QUERIES — a query per compute unit: such a
		// launch is bound by its longest query, and a walker reads its 3 KiB over PCIe once per ~0.4 ms of work.)
		if (!exact && !filter && nq <= DIRECT_IO_MAX_QUERIES && count) {
			const size_t q_bytes = (nq * dim * 4 + 15) & ~size_t(15), key_bytes = nq * k * 8;
			const size_t d_bytes = (nq * k * 4 + 15) & ~size_t(15), c_bytes = (nq * 4 + 15) & ~size_t(15);
			const size_t need = q_bytes + key_bytes + d_bytes + c_bytes;
			if (c.pinned_cap < need) {
				if (c.pinned_io)
					(void)hipHostFree(c.pinned_io);
				c.pinned_io = nullptr, c.pinned_cap = 0;
				HIP_TRY(hipHostMalloc((void **)&c.pinned_io, need, hipHostMallocCoherent));
				c.pinned_cap = need;
			}
			float *pq = reinterpret_cast<float *>(c.pinned_io);
			int64_t *pk = reinterpret_cast<int64_t *>(c.pinned_io + q_bytes);
			float *pd = reinterpret_cast<float *>(c.pinned_io + q_bytes + key_bytes);
			uint32_t *pc = reinterpret_cast<uint32_t *>(c.pinned_io + q_bytes + key_bytes + d_bytes);
			std::memcpy(pq, queries, nq * dim * 4);
			int rc = search_launch(lease.slot, pq, (uint32_t)dim, nq, k, ef, pk, pd, pc, true, nullptr, 0, true);
			if (rc != VSS_OK)
				return rc;
			std::memcpy(out_keys, pk, nq * k * 8);
			if (out_d)
				std::memcpy(out_d, pd, nq * k * 4);
			if (out_counts)
				std::memcpy(out_counts, pc, nq * 4);
			return VSS_OK;
		}
		const uint64_t *d_filter = nullptr;
		if (filter) {
			const uint64_t words = (filter_bits + 63) / 64;
			c.d_filter.ensure(std::max<uint64_t>(words, 1), 0, c.stream);
			HIP_TRY(hipMemcpyAsync(c.d_filter.p, filter, words * 8, hipMemcpyHostToDevice, c.stream));
			d_filter = c.d_filter.p;
		}
		c.d_q.ensure(nq * dim, 0, c.stream);
		c.d_out_keys.ensure(nq * k, 0, c.stream);
		c.d_out_d.ensure(nq * k, 0, c.stream);
		c.d_out_count.ensure(nq, 0, c.stream);
		HIP_TRY(hipMemcpyAsync(c.d_q.p, queries, nq * dim * 4, hipMemcpyHostToDevice, c.stream));
		int rc;
		if (exact) {
			HIP_TRY(hipStreamSynchronize(c.stream)); // the exact kernels run on the index stream
			rc = exact_launch(c.d_q.p, (uint32_t)dim, nq, k, c.d_out_keys.p, c.d_out_d.p, c.d_out_count.p);
		} else {
			rc = search_launch(lease.slot, c.d_q.p, (uint32_t)dim, nq, k, ef, c.d_out_keys.p, c.d_out_d.p, c.d_out_count.p, true,
			                   d_filter, filter_bits);
		}
		if (rc != VSS_OK)
			return rc;
		HIP_TRY(hipMemcpyAsync(out_keys, c.d_out_keys.p, nq * k * 8, hipMemcpyDeviceToHost, c.stream));
		if (out_d)
			HIP_TRY(hipMemcpyAsync(out_d, c.d_out_d.p, nq * k * 4, hipMemcpyDeviceToHost, c.stream));
		if (out_counts)
			HIP_TRY(hipMemcpyAsync(out_counts, c.d_out_count.p, nq * 4, hipMemcpyDeviceToHost, c.stream));
		HIP_TRY(hipStreamSynchronize(c.stream));
		return VSS_OK;
	}

	// ------------------------------------------------------------------ exact search
	int exact_launch(const float *d_queries, uint32_t q_stride, uint64_t nq, uint64_t k, int64_t *d_keys_out,
	                 float *d_dist_out, uint32_t *d_count_out) {
		const uint64_t rows = count;
		if (!nq || !k)
			return VSS_OK;
		std::lock_guard<std::mutex> exact_lock(exact_mu); // score tiles, norms and the index stream are shared
		// running top-(k + 8) per query; exact search is not reachable from the reference's SQL surface (HNSWIndex never passes
		// exact=true), its k is bounded by the select kernel's LDS
		if (k + 8 > SEL_KP_MAX)
			return fail("exact search supports k <= %d", SEL_KP_MAX - 8);
		const uint64_t KP = k + 8;
		if (!rows) {
			HIP_TRY(hipMemsetAsync(d_keys_out, 0xFF, nq * k * 8, stream));
			HIP_TRY(hipMemsetAsync(d_count_out, 0, nq * 4, stream));
			HIP_TRY(hipStreamSynchronize(stream));
			return VSS_OK;
		}
		const uint64_t CH = 32768;
		const uint64_t stride = (uint64_t)V * 4;
		d_row_norm2.ensure(capacity, 0, stream);
		d_qpad.ensure(nq * stride, 0, stream);
		d_q_norm2.ensure(nq, 0, stream);
		d_scores.ensure(nq * CH, 0, stream);
		d_best_s.ensure(nq * KP, 0, stream);
		d_best_i.ensure(nq * KP, 0, stream);
		if (norms_valid_for != mutations) {
			hipLaunchKernelGGL(k_row_norms, dim3(2048), dim3(256), 0, stream,
			                   reinterpret_cast<const float4 *>(d_vectors.p), V, G, logG, (uint32_t)rows, d_row_norm2.p);
			norms_valid_for = mutations;
		}
		HIP_TRY(hipMemsetAsync(d_qpad.p, 0, nq * stride * 4, stream));
		HIP_TRY(hipMemcpy2DAsync(d_qpad.p, stride * 4, d_queries, (size_t)q_stride * 4, dim * 4, nq,
		                         hipMemcpyDeviceToDevice, stream));
		hipLaunchKernelGGL(k_row_norms, dim3(256), dim3(256), 0, stream, reinterpret_cast<const float4 *>(d_qpad.p), V,
		                   G, logG, (uint32_t)nq, d_q_norm2.p);
		// Round 4: from the second chunk on the select is folded into the score tile's epilogue — only scores that beat a
		// query's K'-th best so far are kept (as survivors in a small per-query buffer), and the running top-K' is refreshed
		// from those buffers once per launch of eight chunks' worth of rows instead of from 128 KiB of scores per query after
		// every chunk.  Same answers:
		// a row is dropped only against a threshold that is never below the final one.  A query that collects more survivors
		// than its buffer holds (rows arriving in descending-distance order, say) raises a flag and the search is redone the
		// plain way.  VSS_EXACT_FILTER=0 keeps the plain way throughout (A/B).
		// How many rows a filtered launch may cover follows from the survivors it must expect: its threshold is the K'-th best
		// of the r0 rows seen so far, so a window of w rows in random order leaves about K' * w / r0 of them.  Windows are sized
		// to a quarter of the buffer (w <= r0 * CAND_CAP / (4 K'), at most eight chunks): k = 10 takes eight chunks from the
		// start, k = 200 grows its windows geometrically, and beyond K' = CAND_CAP / 8 (k > 248) — where even one chunk
		// behind one chunk would fill a quarter of the buffer — the plain way is taken from the start (the first
		// window of eight chunks used to overflow almost surely for k above ~248, and the whole pass was then repeated).
		const uint64_t CAND_CAP = SEL_CAP, SELECT_EVERY = 8;
		const bool want_filter = exact_filter && (exact_kernel == 2 || exact_kernel >= 4) && rows > CH && 8 * KP <= CAND_CAP;
		if (want_filter) {
			d_cand_cnt.ensure(nq + 1, 0, stream); // [nq] = the overflow flag
			d_cand_s.ensure(nq * CAND_CAP, 0, stream);
			d_cand_i.ensure(nq * CAND_CAP, 0, stream);
		}
		auto run = [&](bool filtered) {
			HIP_TRY(hipMemsetAsync(d_best_s.p, 0x7F, nq * KP * 4, stream)); // 0x7F7F7F7F = 3.39e38 (acts as +inf)
			HIP_TRY(hipMemsetAsync(d_best_i.p, 0xFF, nq * KP * 4, stream));
			if (filtered)
				HIP_TRY(hipMemsetAsync(d_cand_cnt.p, 0, (nq + 1) * 4, stream));
			// plain chunks are CH rows (the score matrix is nq x CH); a filtered launch stores no scores, so it covers
			// SELECT_EVERY chunks' worth of rows at once and is followed by one select over the survivors it left
			for (uint64_t r0 = 0; r0 < rows;) {
				const bool filter_this = filtered && r0 > 0;
				const uint64_t window = filter_this ? std::min(SELECT_EVERY * CH, std::max(CH, r0 * CAND_CAP / (4 * KP))) : CH;
				const uint64_t r1 = std::min(rows, r0 + window);
				ExactArgs e;
				e.queries = reinterpret_cast<const float4 *>(d_qpad.p);
				e.vectors = reinterpret_cast<const float4 *>(d_vectors.p);
				e.row_norm2 = d_row_norm2.p;
				e.query_norm2 = d_q_norm2.p;
				e.keys = d_keys.p;
				e.V = V;
				e.n_queries = (uint32_t)nq;
				e.row_begin = (uint32_t)r0;
				e.row_end = (uint32_t)r1;
				e.chunk_stride = (uint32_t)(filter_this ? r1 - r0 : CH);
				e.metric = metric;
				e.scores = d_scores.p;
				e.probe = exact_probe;
				e.best_s = d_best_s.p, e.best_i = d_best_i.p, e.KP = (uint32_t)KP, e.cand_cap = (uint32_t)CAND_CAP;
				e.cand_cnt = filter_this ? d_cand_cnt.p : nullptr;
				e.cand_s = d_cand_s.p, e.cand_i = d_cand_i.p;
				if (exact_kernel >= 4) { // round 5: the 128 x 128 tile as persistent workgroups, two per compute unit
					const uint64_t tiles = ((r1 - r0 + 127) / 128) * ((nq + 127) / 128);
					const uint32_t grid = (uint32_t)std::min<uint64_t>(2ull * n_cus, tiles);
					// operands by LDS-DMA (5, the default) where a row is a whole number of 128-byte steps and a tile's rows lie within
					// the 32-bit offsets of the DMA's address form; staged through registers (4) otherwise
					if (exact_kernel == 5 && V % 8 == 0 && 128ull * V * 16 < (1ull << 31)) {
						HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_exact_scores_v4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)X4_LDS_BYTES));
						hipLaunchKernelGGL(k_exact_scores_v4, dim3(grid), dim3(256), X4_LDS_BYTES, stream, e);
					} else {
						const uint32_t lds = X2Shape<2>::LDS_BYTES;
						HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_exact_scores_v3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
						hipLaunchKernelGGL(k_exact_scores_v3, dim3(grid), dim3(256), lds, stream, e);
					}
				} else if (exact_kernel >= 2) { // the software-pipelined tiles (round 3): 128 x 128 (2) or 128 x 256 (3)
					const bool wide = exact_kernel == 3;
					const uint32_t bn = wide ? X2Shape<4>::BN : X2Shape<2>::BN, lds = wide ? X2Shape<4>::LDS_BYTES : X2Shape<2>::LDS_BYTES;
					const void *fn = wide ? reinterpret_cast<const void *>(k_exact_scores_v2<4>) : reinterpret_cast<const void *>(k_exact_scores_v2<2>);
					HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
					dim3 grid((uint32_t)((r1 - r0 + bn - 1) / bn), (uint32_t)((nq + 127) / 128));
					if (wide)
						hipLaunchKernelGGL(k_exact_scores_v2<4>, grid, dim3(256), lds, stream, e);
					else
						hipLaunchKernelGGL(k_exact_scores_v2<2>, grid, dim3(256), lds, stream, e);
				} else {
					dim3 grid((uint32_t)((r1 - r0 + XT_BN - 1) / XT_BN), (uint32_t)((nq + XT_BM - 1) / XT_BM));
					hipLaunchKernelGGL(k_exact_scores, grid, dim3(XT_THREADS), 0, stream, e);
				}
				SelectArgs s;
				s.scores = filter_this ? nullptr : d_scores.p;
				s.chunk_stride = (uint32_t)CH;
				s.chunk_cols = (uint32_t)(r1 - r0);
				s.row_begin = (uint32_t)r0;
				s.KP = (uint32_t)KP;
				s.best_s = d_best_s.p;
				s.best_i = d_best_i.p;
				s.cand_cap = (uint32_t)CAND_CAP;
				s.cand_cnt = d_cand_cnt.p, s.cand_s = d_cand_s.p, s.cand_i = d_cand_i.p;
				s.overflow = d_cand_cnt.p ? d_cand_cnt.p + nq : nullptr;
				hipLaunchKernelGGL(k_exact_select, dim3((uint32_t)nq), dim3(SEL_THREADS), 0, stream, s);
				r0 = r1;
			}
		};
		run(want_filter);
		if (want_filter) {
			uint32_t overflowed = 0;
			HIP_TRY(hipMemcpyAsync(&overflowed, d_cand_cnt.p + nq, 4, hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipStreamSynchronize(stream));
			if (overflowed)
				run(false);
		}
		RerankArgs r;
		r.gv = view();
		r.queries = d_queries;
		r.q_stride = q_stride;
		r.n_queries = (uint32_t)nq;
		r.k = (uint32_t)k;
		r.KP = (uint32_t)KP;
		r.best_i = d_best_i.p;
		r.out_keys = d_keys_out;
		r.out_d = d_dist_out;
		r.out_count = d_count_out;
		const uint32_t lds = align16(V * 16) + 2 * align16((uint32_t)KP * 4);
		launch_by_metric<RerankArgs>(launch_rerank<0>, launch_rerank<1>, launch_rerank<2>, r,
		                             launch_cfg((uint32_t)nq, lds, 64));
		HIP_TRY(hipStreamSynchronize(stream));
		return VSS_OK;
	}

	// ------------------------------------------------------------------ remove
	void ensure_keymap() {
		if (keymap.ready)
			return;
		keymap.init(count + staged);
		for (uint64_t i = 0; i != count + staged; ++i)
			if (keys_h[i] != VSS_FREE_KEY)
				keymap.put(keys_h[i], (uint32_t)i);
	}

	int remove(const int64_t *rowids, uint64_t n, uint64_t *removed) {
		if (removed)
			*removed = 0;
		if (staged || n_pending)
			return fail("cannot remove while staged rows are unlinked (call vss_build_finalize first)");
		if (refuse_while_probing("vss_remove_batch") != VSS_OK)
			return VSS_ERROR;
		ensure_keymap();
		std::vector<uint32_t> slots;
		for (uint64_t i = 0; i != n; ++i) {
			uint32_t slot;
			if (!keymap.find(rowids[i], slot))
				continue;
			if (!free_slots.reserve(free_slots.size() + 1)) // index_dense.hpp:1239-1240
				return fail("Can't allocate memory for a free-list");
			free_slots.push(slot);
			keymap.erase(rowids[i]);
			keys_h[slot] = VSS_FREE_KEY;
			slots.push_back(slot);
		}
		if (!slots.empty()) { // one upload + one kernel for the whole batch
			d_work_build.ensure(slots.size(), 0, stream);
			HIP_TRY(hipMemcpyAsync(d_work_build.p, slots.data(), slots.size() * 4, hipMemcpyHostToDevice, stream));
			const uint32_t tb = 256, gb = (uint32_t)std::min<uint64_t>((slots.size() + tb - 1) / tb, 1024);
			hipLaunchKernelGGL(k_mark_removed, dim3(gb), dim3(tb), 0, stream, d_keys.p, d_work_build.p,
			                   (uint32_t)slots.size());
			HIP_TRY(hipGetLastError());
			HIP_TRY(hipStreamSynchronize(stream));
			tombstones += slots.size();
			mutations++;
		}
		if (removed)
			*removed = slots.size();
		return VSS_OK;
	}

	// ------------------------------------------------------------------ stream format (usearch 2.12)
	uint64_t node_bytes(int level) const {
		return 10 + (4 + 4 * M0) + (uint64_t)level * (4 + 4 * M);
	}
	uint64_t serialized_length() const {
		uint64_t n = 8 + count * dim * 4 + 64 + 40;
		for (uint64_t i = 0; i != count; ++i)
			n += node_bytes(levels_h[i]) + 2;
		return n;
	}

	int save(vss_write_cb write, void *ctx) {
		if (staged || n_pending)
			return fail("cannot serialise with staged, unlinked rows (call vss_build_finalize first)");
		const uint64_t stride = (uint64_t)V * 4;
		auto put = [&](const void *p, uint64_t n) -> bool { return n == 0 || write(ctx, p, n) != 0; };
		uint32_t dims[2] = {(uint32_t)count, (uint32_t)(dim * 4)};
		if (!put(dims, 8))
			return fail("Failed to serialize into stream");
		// vectors in slot order, un-padded, in slabs
		{
			const uint64_t slab = std::max<uint64_t>(1, (64ull << 20) / (dim * 4));
			std::vector<float> buf(slab * dim);
			for (uint64_t r0 = 0; r0 < count; r0 += slab) {
				const uint64_t nr = std::min(slab, count - r0);
				HIP_TRY(hipMemcpy2DAsync(buf.data(), dim * 4, d_vectors.p + r0 * stride, stride * 4, dim * 4, nr,
				                         hipMemcpyDeviceToHost, stream));
				HIP_TRY(hipStreamSynchronize(stream));
				if (!put(buf.data(), nr * dim * 4))
					return fail("Failed to serialize into stream");
			}
		}
		uint8_t head[64];
		std::memset(head, 0, 64);
		std::memcpy(head, "usearch", 7);
		uint16_t ver[3] = {2, 12, 0};
		std::memcpy(head + 7, ver, 6);
		head[13] = metric == 0 ? 'e' : metric == 1 ? 'c' : 'i';
		head[14] = 11, head[15] = 20, head[16] = 15;
		uint64_t present = count - tombstones, deleted = tombstones, dimensions = dim;
		std::memcpy(head + 17, &present, 8);
		std::memcpy(head + 25, &deleted, 8);
		std::memcpy(head + 33, &dimensions, 8);
		if (!put(head, 64))
			return fail("Failed to serialize into stream");
		uint64_t gh[5] = {count, M, M0, (uint64_t)(int64_t)(count ? max_level : -1), entry};
		if (!put(gh, 40))
			return fail("Failed to serialize the header into stream");
		std::vector<int16_t> lv(count);
		for (uint64_t i = 0; i != count; ++i)
			lv[i] = levels_h[i];
		if (!put(lv.data(), count * 2))
			return fail("Failed to serialize into stream");
		std::vector<uint32_t> l0(count * M0), lu(n_upper * M);
		HIP_TRY(hipMemcpyAsync(l0.data(), d_links0.p, l0.size() * 4, hipMemcpyDeviceToHost, stream));
		if (n_upper)
			HIP_TRY(hipMemcpyAsync(lu.data(), d_links_up.p, lu.size() * 4, hipMemcpyDeviceToHost, stream));
		HIP_TRY(hipStreamSynchronize(stream));
		std::vector<uint8_t> out;
		out.reserve(1 << 20);
		auto emit_list = [&](const uint32_t *src, uint64_t cap) {
			uint32_t cnt = 0;
			while (cnt < cap && src[cnt] != EMPTY_SLOT)
				cnt++;
			const size_t at = out.size();
			out.resize(at + 4 + 4 * cap, 0);
			std::memcpy(out.data() + at, &cnt, 4);
			std::memcpy(out.data() + at + 4, src, 4 * (size_t)cnt);
		};
		for (uint64_t i = 0; i != count; ++i) {
			const size_t at = out.size();
			out.resize(at + 10);
			std::memcpy(out.data() + at, &keys_h[i], 8);
			std::memcpy(out.data() + at + 8, &lv[i], 2);
			emit_list(l0.data() + i * M0, M0);
			for (int l = 0; l < lv[i]; ++l)
				emit_list(lu.data() + ((uint64_t)upper_off_h[i] + l) * M, M);
			if (out.size() >= (1 << 20) || i + 1 == count) {
				if (!put(out.data(), out.size()))
					return fail("Failed to serialize into stream");
				out.clear();
			}
		}
		return VSS_OK;
	}

	int load(vss_read_cb read, void *ctx) {
		if (refuse_while_probing("vss_load") != VSS_OK)
			return VSS_ERROR;
		auto get = [&](void *p, uint64_t n) -> bool { return n == 0 || read(ctx, p, n) != 0; };
		uint32_t dims[2];
		if (!get(dims, 8))
			return fail("Failed to read 32-bit dimensions of the matrix");
		const uint64_t rows = dims[0], cols = dims[1];
		std::vector<float> vecs(rows * cols / 4);
		if (!get(vecs.data(), rows * cols))
			return fail("Failed to read vectors");
		uint8_t head[64];
		if (!get(head, 64))
			return fail("Failed to read the index ");
		if (std::memcmp(head, "usearch", 7) != 0)
			return fail("Magic header mismatch - the file isn't an index");
		uint16_t ver_major;
		std::memcpy(&ver_major, head + 7, 2);
		if (ver_major != 2)
			return fail("File format may be different, please rebuild");
		if (head[15] != 20)
			return fail("Key type doesn't match, consider rebuilding");
		if (head[16] != 15)
			return fail("Slot type doesn't match, consider rebuilding");
		if (head[14] != 11)
			return fail("Only f32 vectors are supported");
		uint64_t dimensions;
		std::memcpy(&dimensions, head + 33, 8);
		const int m = head[13] == 'e' ? 0 : head[13] == 'c' ? 1 : head[13] == 'i' ? 2 : -1;
		if (m < 0)
			return fail("Unsupported metric kind in stream");
		uint64_t gh[5];
		if (!get(gh, 40))
			return fail("Failed to pull the header from the stream");
		if (rows && cols != dimensions * 4)
			return fail("Vector size in stream doesn't match its dimensions");
		// Parse and validate the whole stream into temporaries first: a corrupt or truncated stream must leave the handle
		// as it was, and every slot number the kernels will follow must be in range.
		const uint64_t sM = gh[1], sM0 = gh[2];
		if (rows != gh[0])
			return fail("Index size and the number of vectors doesn't match");
		if (rows && (sM < 2 || sM0 < sM))
			return fail("Corrupt connectivity in stream");
		if (rows >= 0x7FFFFFFFull)
			return fail("capacity above 2^31-1 slots is not supported");
		std::vector<int16_t> lv(rows);
		std::vector<uint8_t> t_levels(rows);
		std::vector<uint32_t> t_off(rows), t_owner, l0, lu;
		std::vector<int64_t> t_keys(rows);
		std::vector<uint32_t> named_by(rows, EMPTY_SLOT); // last list (numbered as read) that named each slot
		uint32_t lists_read = 0;
		bool t_repeat = false;
		uint64_t upper = 0, t_tomb = 0;
		if (rows) {
			if (!get(lv.data(), rows * 2))
				return fail("Failed to pull nodes levels from the stream");
			for (uint64_t i = 0; i != rows; ++i) {
				if (lv[i] < 0 || lv[i] > 255)
					return fail("Corrupt level in stream");
				t_levels[i] = (uint8_t)lv[i];
				t_off[i] = (uint32_t)upper;
				upper += lv[i];
			}
			const int64_t s_max_level = (int64_t)gh[3];
			if (gh[4] >= rows || s_max_level != lv[gh[4]])
				return fail("Corrupt entry point in stream");
			for (uint64_t i = 0; i != rows; ++i)
				if (lv[i] > s_max_level)
					return fail("Corrupt level in stream");
			t_owner.resize(upper);
			l0.assign(rows * sM0, EMPTY_SLOT);
			lu.assign(upper * sM, EMPTY_SLOT);
			std::vector<uint8_t> tape;
			for (uint64_t i = 0; i != rows; ++i) {
				tape.resize(10 + (4 + 4 * sM0) + (uint64_t)lv[i] * (4 + 4 * sM));
				if (!get(tape.data(), tape.size()))
					return fail("Failed to pull nodes from the stream");
				std::memcpy(&t_keys[i], tape.data(), 8);
				int16_t record_level;
				std::memcpy(&record_level, tape.data() + 8, 2);
				if (record_level != lv[i])
					return fail("Corrupt level in stream");
				t_tomb += t_keys[i] == VSS_FREE_KEY;
				const uint8_t *p = tape.data() + 10;
				for (int l = 0; l <= lv[i]; ++l) {
					const uint64_t cap = l ? sM : sM0;
					uint32_t cnt;
					std::memcpy(&cnt, p, 4);
					if (cnt > cap)
						return fail("Corrupt neighbour count in stream");
					uint32_t *dst = l ? lu.data() + ((uint64_t)t_off[i] + l - 1) * sM : l0.data() + i * sM0;
					std::memcpy(dst, p + 4, 4 * (size_t)cnt);
					for (uint32_t j = 0; j != cnt; ++j) {
						if (dst[j] >= rows || lv[dst[j]] < l)
							return fail("Corrupt neighbour slot in stream");
						t_repeat |= named_by[dst[j]] == lists_read;
						named_by[dst[j]] = lists_read;
					}
					lists_read++;
					if (l)
						t_owner[t_off[i] + l - 1] = (uint32_t)i;
					p += 4 + 4 * cap;
				}
			}
		}
		// adopt the stream's shape (usearch load_from_stream resets and re-reserves: index.hpp:3172-3186)
		release_graph_only();
		reset_graph();
		configure(dimensions, m, sM, sM0);
		if (!rows)
			return VSS_OK;
		int rc = reserve(rows, 1);
		if (rc != VSS_OK)
			return rc;
		std::memcpy(levels_h.data(), t_levels.data(), rows);
		std::memcpy(upper_off_h.data(), t_off.data(), rows * 4);
		std::memcpy(keys_h.data(), t_keys.data(), rows * 8);
		ensure_upper(upper);
		list_owner_h.swap(t_owner);
		tombstones = t_tomb;
		lists_may_repeat = t_repeat;
		const uint64_t stride = (uint64_t)V * 4;
		HIP_TRY(hipMemcpy2DAsync(d_vectors.p, stride * 4, vecs.data(), dim * 4, dim * 4, rows, hipMemcpyHostToDevice,
		                         stream));
		HIP_TRY(hipMemcpyAsync(d_links0.p, l0.data(), l0.size() * 4, hipMemcpyHostToDevice, stream));
		if (upper) {
			HIP_TRY(hipMemcpyAsync(d_links_up.p, lu.data(), lu.size() * 4, hipMemcpyHostToDevice, stream));
			HIP_TRY(hipMemcpyAsync(d_list_owner.p, list_owner_h.data(), upper * 4, hipMemcpyHostToDevice, stream));
		}
		HIP_TRY(hipMemcpyAsync(d_levels.p, levels_h.data(), rows, hipMemcpyHostToDevice, stream));
		HIP_TRY(hipMemcpyAsync(d_upper_off.p, upper_off_h.data(), rows * 4, hipMemcpyHostToDevice, stream));
		HIP_TRY(hipMemcpyAsync(d_keys.p, keys_h.data(), rows * 8, hipMemcpyHostToDevice, stream));
		HIP_TRY(hipStreamSynchronize(stream));
		n_upper = upper;
		count = rows;
		max_level = (int)(int16_t)gh[3];
		entry = (uint32_t)gh[4];
		if (tombstones) { // reindex_keys_ rebuilds the free list in slot order (index_dense.hpp:1901-1930)
			free_slots.clear();
			free_slots.reserve(tombstones);
			for (uint64_t i = 0; i != rows; ++i)
				if (keys_h[i] == VSS_FREE_KEY)
					free_slots.push((uint32_t)i);
		}
		mutations++;
		return VSS_OK;
	}

	void release_graph_only() {
		d_vectors.free(), d_links0.free(), d_links_up.free(), d_upper_off.free(), d_list_owner.free();
		d_levels.free(), d_keys.free(), d_list_count.free(), d_list_offset.free(), d_row_norm2.free();
	}

	void configure(uint64_t dim_, int metric_, uint64_t M_, uint64_t M0_) {
		dim = dim_;
		metric = metric_;
		M = M_;
		M0 = M0_;
		V = (uint32_t)((dim + 3) / 4);
		G = (uint32_t)std::min<size_t>(64, ceil_pow2(V));
		logG = log2u(G);
		inv_log_m = inverse_log_connectivity(M);
	}

	// ------------------------------------------------------------------ compact (drops tombstones)
	int compact(bool reorder);
	bool last_compact_reordered = false;
	bool reorder_after_build = false;

	void level_stats(uint64_t level, uint64_t *out4) {
		// usearch index.hpp:3010-3027 (including its inverted max_edges connectivity)
		uint64_t nodes = 0, edges = 0;
		std::vector<uint32_t> lists;
		if (level == 0) {
			lists.resize(count * M0);
			if (count)
				HIP_TRY(hipMemcpy(lists.data(), d_links0.p, lists.size() * 4, hipMemcpyDeviceToHost));
			for (uint64_t i = 0; i != count; ++i) {
				nodes++;
				for (uint64_t j = 0; j != M0 && lists[i * M0 + j] != EMPTY_SLOT; ++j)
					edges++;
			}
		} else {
			lists.resize(n_upper * M);
			if (n_upper)
				HIP_TRY(hipMemcpy(lists.data(), d_links_up.p, lists.size() * 4, hipMemcpyDeviceToHost));
			for (uint64_t i = 0; i != count; ++i) {
				if (levels_h[i] < level)
					continue;
				nodes++;
				const uint32_t *lp = lists.data() + ((uint64_t)upper_off_h[i] + level - 1) * M;
				for (uint64_t j = 0; j != M && lp[j] != EMPTY_SLOT; ++j)
					edges++;
			}
		}
		out4[0] = nodes;
		out4[1] = edges;
		out4[2] = nodes * (level ? M0 : M);
		out4[3] = nodes * (10 + 4 + 4 * (level ? M : M0));
	}
};

// ---------------------------------------------------------------------------------------------------------
// The launch gate is per DEVICE, not per index: launches of different indexes on one GPU (row-range shards placed on the
// same device, host/sharded_index.hpp) compete for the same compute units exactly like launches of one index.
// g_gate[device] names the context of the most recent gated launch.  Every device slot has a mutex of its own, held while
// waiting — which also keeps the named index alive: vss_destroy clears its entry under the same mutex — so a wait on one
// GPU never delays launches (or vss_destroy) on another one of a process that drives several.
// ---------------------------------------------------------------------------------------------------------
namespace {
struct GateEntry {
	vss_index *owner = nullptr;
	int ctx = -1;
};
struct DeviceGate {
	std::mutex mu;
	GateEntry last;
};
DeviceGate g_gate[64];
} // namespace

void vss_index::wait_for_drain_of_previous_launch(int slot) {
	DeviceGate &dg = g_gate[(unsigned)device % 64];
	std::lock_guard<std::mutex> lk(dg.mu);
	GateEntry &g = dg.last;
	const GateEntry prev = g;
	g.owner = this, g.ctx = slot;
	if (!search_gating || !prev.owner || (prev.owner == this && prev.ctx == slot))
		return;
	SearchCtx &pc = prev.owner->ctx[prev.ctx];
	if (!pc.h_queue || !pc.ev1)
		return;
	volatile uint32_t *drained = pc.h_queue + 2;
	while (*drained == 0) {
		if (hipEventQuery(pc.ev1) != hipErrorNotReady)
			break; // finished (or never started): nothing to wait for
		std::this_thread::yield();
	}
	(void)hipGetLastError(); // hipErrorNotReady is an answer, not an error to be found by the next launch check
}

// ---------------------------------------------------------------------------------------------------------
// compact.  Two things happen, both on the device:
//   * tombstoned nodes are dropped and links to them removed, the remaining links keep their order (the DOCUMENTED
//     behaviour of PRAGMA hnsw_compact_index, reference README.md:69; usearch's own compact keeps them);
//   * reorder = true (vss_compact): the survivors are renumbered in the reference's compaction order — (level descending,
//     cluster ascending), cluster = the node the greedy descent from the entry lands on above level 0 (index_gt::compact,
//     index.hpp:3405-3494) — with the old slot as the final key (the reference's std::sort leaves that to the library).
//     Nodes of one cluster become neighbours in memory: the rows a search touches lie in a few contiguous runs.
//     reorder = false (vss_compact_ex(.., 0), and the fall-back when a second vector buffer cannot be allocated): the
//     survivors keep their relative order.
// The host derives the slot maps from its key / level mirrors and the cluster array (O(n) counting sorts) and uploads them;
// lists, keys, levels and the vector rows move on the device (k_node_clusters, k_compact_links, k_compact_rows).  Every
// new array is built aside and swapped in only when everything succeeded.  Mirrored by the oracle's compact_dropping() /
// compact_reordering().
// ---------------------------------------------------------------------------------------------------------
int vss_index::compact(bool reorder) {
	if (staged || n_pending)
		return fail("cannot compact with staged, unlinked rows");
	if (refuse_while_probing("vss_compact") != VSS_OK)
		return VSS_ERROR;
	last_compact_reordered = false; // true only once reordered arrays have been swapped in (vss_compact_ex's out_reordered)
	if (!count || (!tombstones && !reorder))
		return VSS_OK;
	const uint64_t stride = (uint64_t)V * 4;
	const uint64_t live = count - tombstones;
	// The rows are gathered into a second buffer that replaces the old one on success.  Without one (not enough free HBM) a
	// permutation is impossible: the compaction then only prunes, moving the rows down in place.
	DevBuf<float> n_vectors;
	struct FreeOnExit { // whatever way this function is left (the host-side sorts below allocate and may throw): no HBM stays behind
		DevBuf<float> &b;
		~FreeOnExit() {
			b.free();
		}
	} n_vectors_guard {n_vectors};
	{
		float *np = nullptr;
		const char *in_place = getenv("VSS_COMPACT_IN_PLACE"); // tests: take the no-second-buffer path although HBM is free
		if (!(in_place && atoi(in_place)) && hipMalloc(&np, d_vectors.n * sizeof(float)) == hipSuccess) {
			n_vectors.p = np, n_vectors.n = d_vectors.n;
		} else {
			(void)hipGetLastError();
			reorder = false;
			if (!tombstones)
				return VSS_OK;
		}
	}
	const bool fresh_rows = n_vectors.p != nullptr;
	// ---- the order of the survivors: src_of[new slot] = old slot
	std::vector<uint32_t> src_of;
	src_of.reserve(live);
	if (!reorder) {
		for (uint64_t i = 0; i != count; ++i)
			if (keys_h[i] != VSS_FREE_KEY)
				src_of.push_back((uint32_t)i);
	} else {
		std::vector<uint32_t> cluster(count);
		DevBuf<uint32_t> d_cluster;
		try {
			d_cluster.ensure(count, 0, stream);
			d_work_stats.ensure(4, 0, stream, 0);
			ClusterArgs ca;
			ca.gv = view();
			ca.count = (uint32_t)count;
			ca.entry = entry;
			ca.max_level = max_level;
			ca.list_cap_max = list_cap_max();
			ca.cluster = d_cluster.p;
			ca.work_stats = nullptr;
			const uint32_t lds = wave_lds_bytes(4, V, ca.list_cap_max, 16);
			launch_by_metric<ClusterArgs>(launch_clusters<0>, launch_clusters<1>, launch_clusters<2>, ca,
			                              launch_cfg((uint32_t)std::min<uint64_t>(count, (uint64_t)n_cus * 64), lds, 64));
			HIP_TRY(hipMemcpyAsync(cluster.data(), d_cluster.p, count * 4, hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipStreamSynchronize(stream));
		} catch (...) {
			d_cluster.free(), n_vectors.free();
			throw;
		}
		d_cluster.free();
		// stable counting sorts, least significant key first: by cluster (ascending), then by level (descending); the input
		// is in slot order, so equal (level, cluster) keep ascending old slots
		std::vector<uint32_t> by_cluster(live), start(count + 1, 0);
		for (uint64_t i = 0; i != count; ++i)
			if (keys_h[i] != VSS_FREE_KEY)
				start[cluster[i] + 1]++;
		for (uint64_t c = 0; c != count; ++c)
			start[c + 1] += start[c];
		for (uint64_t i = 0; i != count; ++i)
			if (keys_h[i] != VSS_FREE_KEY)
				by_cluster[start[cluster[i]]++] = (uint32_t)i;
		uint64_t lstart[257] = {0};
		for (uint32_t sidx : by_cluster)
			lstart[(255 - levels_h[sidx]) + 1]++;
		for (int b = 0; b != 256; ++b)
			lstart[b + 1] += lstart[b];
		src_of.resize(live);
		for (uint32_t sidx : by_cluster)
			src_of[lstart[255 - levels_h[sidx]]++] = sidx;
	}
	std::vector<uint32_t> remap(count, EMPTY_SLOT), noff(live);
	uint64_t nup = 0, first_moved = live;
	for (uint64_t t = 0; t != live; ++t) {
		remap[src_of[t]] = (uint32_t)t;
		noff[t] = (uint32_t)nup;
		nup += levels_h[src_of[t]];
		if (src_of[t] != t)
			first_moved = std::min(first_moved, t);
	}
	// new entry point: the old one if it survives, else the surviving node of the highest level (lowest new slot among
	// equals: with the reordering that is new slot 0)
	int nml = -1;
	uint32_t nentry = 0;
	if (remap[entry] != EMPTY_SLOT) {
		nml = max_level;
		nentry = remap[entry];
	} else {
		for (uint64_t t = 0; t != live; ++t)
			if ((int)levels_h[src_of[t]] > nml)
				nml = levels_h[src_of[t]], nentry = (uint32_t)t;
	}
	DevBuf<uint32_t> d_remap, d_src, d_noff, n_links0, n_links_up, n_owner, n_upper_off;
	DevBuf<uint8_t> n_levels;
	DevBuf<int64_t> n_keys;
	DevBuf<float> staging;
	auto drop_scratch = [&] {
		d_remap.free(), d_src.free(), d_noff.free(), n_links0.free(), n_links_up.free(), n_owner.free(), n_upper_off.free();
		n_levels.free(), n_keys.free(), staging.free(), n_vectors.free();
	};
	try {
		d_remap.ensure(count, 0, stream), d_src.ensure(std::max<uint64_t>(live, 1), 0, stream);
		d_noff.ensure(std::max<uint64_t>(live, 1), 0, stream);
		HIP_TRY(hipMemcpyAsync(d_remap.p, remap.data(), count * 4, hipMemcpyHostToDevice, stream));
		HIP_TRY(hipMemcpyAsync(d_src.p, src_of.data(), live * 4, hipMemcpyHostToDevice, stream));
		HIP_TRY(hipMemcpyAsync(d_noff.p, noff.data(), live * 4, hipMemcpyHostToDevice, stream));
		n_links0.ensure(d_links0.n, 0, stream, 0xFF);
		n_links_up.ensure(d_links_up.n, 0, stream, 0xFF);
		n_owner.ensure(d_list_owner.n, 0, stream, 0);
		n_upper_off.ensure(d_upper_off.n, 0, stream, 0);
		n_levels.ensure(d_levels.n, 0, stream, 0);
		n_keys.ensure(d_keys.n, 0, stream, 0);
		HIP_TRY(hipMemcpyAsync(n_upper_off.p, noff.data(), live * 4, hipMemcpyHostToDevice, stream));
		CompactArgs a;
		a.src_of = d_src.p, a.remap = d_remap.p, a.live = (uint32_t)live;
		a.V = V, a.M = (uint32_t)M, a.M0 = (uint32_t)M0;
		a.vectors = reinterpret_cast<const float4 *>(d_vectors.p);
		a.links0 = d_links0.p, a.links_up = d_links_up.p, a.upper_off = d_upper_off.p;
		a.levels = d_levels.p, a.keys = d_keys.p;
		a.links0_new = n_links0.p, a.links_up_new = n_links_up.p, a.list_owner_new = n_owner.p;
		a.upper_off_new = d_noff.p, a.levels_new = n_levels.p, a.keys_new = n_keys.p;
		a.staging = nullptr;
		if (live) {
			const uint32_t gl = (uint32_t)std::min<uint64_t>((live + 3) / 4, 16384);
			hipLaunchKernelGGL(k_compact_links, dim3(gl), dim3(256), 0, stream, a);
			HIP_TRY(hipGetLastError());
		}
		if (fresh_rows) { // every surviving row is gathered into the fresh buffer (a permutation when reordering)
			HIP_TRY(hipMemsetAsync(n_vectors.p + live * stride, 0, (n_vectors.n - live * stride) * sizeof(float), stream));
			a.staging = reinterpret_cast<float4 *>(n_vectors.p);
			for (uint64_t t0 = 0; t0 < live; t0 += 1u << 30) { // (k_compact_rows numbers rows with 32 bits)
				const uint64_t n = std::min<uint64_t>(1u << 30, live - t0);
				a.staging = reinterpret_cast<float4 *>(n_vectors.p + t0 * stride);
				hipLaunchKernelGGL(k_compact_rows, dim3((uint32_t)std::min<uint64_t>((n + 3) / 4, 16384)), dim3(256), 0, stream, a,
				                   (uint32_t)t0, (uint32_t)n);
			}
			HIP_TRY(hipGetLastError());
		} else {
			// no second buffer (and therefore order kept): everything below the first tombstone stays; the rest moves DOWN
			// chunk by chunk through a staging buffer (a chunk's sources never lie below its destinations).  In place:
			// d_vectors is only rewritten here, after every other new array has been built; an error in this loop leaves rows
			// moved but the graph arrays untouched — vss_compact then reports the error and the index must be reloaded
			// (stated in vssgpu.h).
			const uint64_t chunk_rows = std::max<uint64_t>(1, std::min<uint64_t>((1ull << 30) / (stride * 4), live));
			HIP_TRY(hipStreamSynchronize(stream)); // links, keys and levels are complete before the first row moves
			if (first_moved < live) {
				staging.ensure(chunk_rows * stride, 0, stream);
				a.staging = reinterpret_cast<float4 *>(staging.p);
				for (uint64_t t0 = first_moved; t0 < live; t0 += chunk_rows) {
					const uint64_t n = std::min(chunk_rows, live - t0);
					const uint32_t gr = (uint32_t)std::min<uint64_t>((n + 3) / 4, 8192);
					hipLaunchKernelGGL(k_compact_rows, dim3(gr), dim3(256), 0, stream, a, (uint32_t)t0, (uint32_t)n);
					HIP_TRY(hipMemcpyAsync(d_vectors.p + t0 * stride, staging.p, n * stride * 4, hipMemcpyDeviceToDevice, stream));
				}
				HIP_TRY(hipGetLastError());
			}
			HIP_TRY(hipMemsetAsync(d_vectors.p + live * stride, 0, (count - live) * stride * 4, stream));
		}
		HIP_TRY(hipStreamSynchronize(stream));
	} catch (...) {
		drop_scratch();
		throw;
	}
	std::swap(d_links0, n_links0), std::swap(d_links_up, n_links_up), std::swap(d_list_owner, n_owner);
	std::swap(d_upper_off, n_upper_off), std::swap(d_levels, n_levels), std::swap(d_keys, n_keys);
	if (fresh_rows)
		std::swap(d_vectors, n_vectors);
	last_compact_reordered = reorder;
	drop_scratch();
	// host mirrors
	{
		std::vector<int64_t> nk(keys_h.size(), 0);
		std::vector<uint8_t> nl(levels_h.size(), 0);
		std::vector<uint32_t> no(upper_off_h.size(), 0);
		list_owner_h.assign(nup, 0);
		for (uint64_t t = 0; t != live; ++t) {
			const uint32_t sidx = src_of[t];
			nk[t] = keys_h[sidx];
			const uint8_t lv = levels_h[sidx];
			nl[t] = lv;
			no[t] = noff[t];
			for (int l = 0; l < lv; ++l)
				list_owner_h[noff[t] + l] = (uint32_t)t;
		}
		keys_h.swap(nk), levels_h.swap(nl), upper_off_h.swap(no);
	}
	count = live;
	n_upper = nup;
	tombstones = 0;
	free_slots.clear();
	max_level = live ? nml : -1;
	entry = nentry;
	keymap = KeyMap();
	mutations++;
	return VSS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------
#define VSS_GUARD_WITH(LOCK, index, ...)                                                                               \
	if (!(index))                                                                                                      \
		return VSS_ERROR;                                                                                              \
	LOCK _lk((index)->rw);                                                                                             \
	try {                                                                                                              \
		(void)hipSetDevice((index)->device);                                                                           \
		__VA_ARGS__                                                                                                    \
	} catch (const HipError &e) {                                                                                      \
		return (index)->fail("HIP error %d (%s) in %s", (int)e.code, hipGetErrorString(e.code), e.what);               \
	} catch (const std::exception &e) {                                                                                \
		return (index)->fail("%s", e.what());                                                                          \
	}
// calls that change the graph or its buffers: exclusive
#define VSS_GUARD(index, ...) VSS_GUARD_WITH(std::unique_lock<std::shared_mutex>, index, __VA_ARGS__)
// searches and read-only queries: any number at once
#define VSS_SHARED(index, ...) VSS_GUARD_WITH(std::shared_lock<std::shared_mutex>, index, __VA_ARGS__)

// ---- tuning options (vss_set_option; no reference counterpart: results never depend on any of them) --------------------------
// One entry per knob.  Round 6: rounds 2-5 exported a setter per knob — ten vss_set_search_* functions a
// DuckDB maintainer had to read past; they are this table now, and the environment variables of the same knobs (read once, in
// vss_create) go through the same checks: a value out of range is REFUSED and reported instead of being clamped silently.
namespace {
struct OptionRule {
	const char *name;
	int64_t lo, hi;
	void (*apply)(vss_index *, int64_t);
	const char *what;
};
const OptionRule OPTION_RULES[] = {
    {"search.waves", 2, 16, [](vss_index *h, int64_t v) { h->search_waves = (uint32_t)v; },
     "wavefronts per workgroup of the search engine"},
    {"search.walkers", 0, ENGINE_MAX_WALKERS, [](vss_index *h, int64_t v) { h->search_walkers = (uint32_t)v; },
     "walking waves among them (0 = chosen per launch); at least one scoring wave must remain"},
    {"search.solo", 0, 2, [](vss_index *h, int64_t v) { h->search_solo = (uint32_t)v; },
     "the one-wave-per-query shape: 0 never, 1 automatic, 2 always"},
    {"search.solo_max_queries", 1, 1 << 30, [](vss_index *h, int64_t v) { h->solo_max_queries = (uint32_t)v; },
     "largest launch the automatic rule gives the solo shape"},
    {"search.team", 0, 1, [](vss_index *h, int64_t v) { h->search_team = v != 0; }, "eight-wave teams in the solo shape"},
    {"search.crew", 0, 31,
     [](vss_index *h, int64_t v) {
	     h->search_crew = (v & 1) != 0;
	     if (v & 16) // explicit refinements (A/B measurements): bits 2-3 = CREW_SPARE_SIMD | CREW_NO_REQUESTS
		     h->search_crew_tune = (uint32_t)v & (CREW_SPARE_SIMD | CREW_NO_REQUESTS);
     },
     "the last walker of a workgroup runs its scoring waves as a crew (1 | 16 | refinement bits 4, 8)"},
    {"search.pipelined", 0, 1, [](vss_index *h, int64_t v) { h->search_pipelined = v != 0; }, "software-pipelined level search"},
    {"search.wide_lists", 0, 1, [](vss_index *h, int64_t v) { h->search_wide_lists = v != 0; },
     "limits of 257-512 in 12-wave workgroups with the pipelined level search"},
    {"search.visited_compact", 0, 1, [](vss_index *h, int64_t v) { h->visited_compact_on = v != 0; },
     "compact exact visited sets (16-bit cells) in LDS at limits of 257-512"},
    {"search.visited_lds_log2_max", 0, 14, [](vss_index *h, int64_t v) { h->hash_lds_max_override = (uint32_t)v; },
     "largest visited-set table (log2 of its 32-bit cells) kept in LDS with several walkers per workgroup; 0 = the engine's own "
     "(13); 1 = every such table in HBM"},
    {"search.visited_cells_per_limit", 0, 1 << 20, [](vss_index *h, int64_t v) { h->visited_per_limit = (uint64_t)v; },
     "visited-set cells per entry of the search limit; 0 = the sizing rule's own, else at least 4"},
    {"search.retry_in_place", 0, 1, [](vss_index *h, int64_t v) { h->retry_in_place = v != 0; },
     "a query that outgrows its LDS-resident visited set is repeated by its walker within the launch"},
    {"search.probe_flag_wait", 0, 1, [](vss_index *h, int64_t v) { h->probe_flag_wait = v != 0; },
     "host-pointer probes of at most 256 queries wait on a pinned flag instead of the stream"},
    {"search.lookahead", 0, ENGINE_MAX_WALKERS, [](vss_index *h, int64_t v) { h->search_spec_active = (uint32_t)v; },
     "one expansion of look-ahead while at most this many walkers of a workgroup still run (0 = off)"},
    {"search.gating", 0, 1, [](vss_index *h, int64_t v) { h->search_gating = v != 0; },
     "a launch is issued when its predecessor on the device starts to drain"},
};
// validated assignment (the caller holds the index exclusively, or is vss_create); nullptr = done, else why not
const char *set_option_checked(vss_index *h, const char *name, int64_t value) {
	for (const OptionRule &r : OPTION_RULES) {
		if (std::strcmp(r.name, name))
			continue;
		if (value < r.lo || value > r.hi)
			return "value out of range";
		if (!std::strcmp(name, "search.visited_cells_per_limit") && value != 0 && value < 4)
			return "value out of range";
		const uint32_t waves = !std::strcmp(name, "search.waves") ? (uint32_t)value : h->search_waves;
		const uint32_t walkers = !std::strcmp(name, "search.walkers") ? (uint32_t)value : h->search_walkers;
		if (walkers && walkers >= waves)
			return "at least one scoring wave must remain";
		r.apply(h, value);
		return nullptr;
	}
	return "no such option";
}
} // namespace

extern "C" {

const char *vss_version(void) {
	return "vssgpu 0.1 (gfx950, wave64)";
}

int vss_create(uint64_t dim, int metric, uint64_t M, uint64_t M0, uint64_t efc, uint64_t efs, int device,
               vss_index **out) {
	if (!out)
		return VSS_ERROR;
	*out = nullptr;
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= device || device < 0) {
		fprintf(stderr, "vssgpu: no HIP device %d available (found %d) — the engine has no CPU fallback\n", device,
		        n_dev);
		return VSS_ERROR;
	}
	if (!dim || metric < 0 || metric > 2 || M < 2 || M0 < 2)
		return VSS_ERROR;
	if (M0 < M) {
		// usearch gives a new node up to M links on EVERY level, level 0 included (index.hpp:3665), and stores them in the
		// M0 cells of its base list: with M0 < M the reference itself writes past the list (heap corruption, observed with
		// the reference build).  There is nothing to be compatible with; refuse.
		fprintf(stderr, "vssgpu: M0 (%llu) must not be smaller than M (%llu)\n", (unsigned long long)M0,
		        (unsigned long long)M);
		return VSS_ERROR;
	}
	auto *h = new vss_index();
	h->device = device;
	h->configure(dim, metric, M, M0);
	h->efc = efc ? efc : 128;
	h->efs = efs ? efs : 64;
	if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
		delete h;
		return VSS_ERROR;
	}
	h->own_stream = true;
#ifdef VSS_PARANOID
	if (hipHostMalloc((void **)&h->h_debug, 128 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
		vss_destroy(h);
		return VSS_ERROR;
	}
	std::memset(h->h_debug, 0, 128 * sizeof(uint32_t));
#endif
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
		h->n_cus = (uint32_t)prop.multiProcessorCount;
	// A/B knobs from the environment (debugging and measurement sessions: tools/README.md).  Those that are options go through
	// vss_set_option's checks — a value out of range is refused and reported, never clamped silently.  Read ONCE,
	// here: setting them after vss_create has no effect on this index.
	static const struct { const char *env, *option; } ENV_OPTIONS[] = {
	    {"VSS_SEARCH_WAVES", "search.waves"}, {"VSS_SEARCH_WALKERS", "search.walkers"}, {"VSS_SEARCH_SPEC", "search.lookahead"},
	    {"VSS_SEARCH_SOLO", "search.solo"}, {"VSS_SEARCH_SOLO_MAX", "search.solo_max_queries"}, {"VSS_SEARCH_TEAM", "search.team"},
	    {"VSS_SEARCH_CREW", "search.crew"}, {"VSS_SEARCH_PIPELINED", "search.pipelined"},
	    {"VSS_SEARCH_WIDE_LISTS", "search.wide_lists"}, {"VSS_VISITED_COMPACT", "search.visited_compact"},
	    {"VSS_SEARCH_RETRY_IN_PLACE", "search.retry_in_place"}, {"VSS_HASH_LDS_MAX_LOG2", "search.visited_lds_log2_max"},
	    {"VSS_VISITED_PER_LIMIT", "search.visited_cells_per_limit"}, {"VSS_PROBE_FLAG_WAIT", "search.probe_flag_wait"},
	};
	for (const auto &e : ENV_OPTIONS)
		if (const char *t = getenv(e.env)) {
			if (const char *why = set_option_checked(h, e.option, (int64_t)atoll(t)))
				fprintf(stderr, "vssgpu: %s=%s ignored (%s: %s)\n", e.env, t, e.option, why);
		}
	// debug-only knobs without an option (kernel selection, probes)
	if (const char *t = getenv("VSS_SEARCH_REG_QUEUE"))
		h->search_reg_queue = atoi(t) != 0;
	if (const char *t = getenv("VSS_SEARCH_REG_QUEUE_MAX"))
		h->reg_queue_max_limit = (uint32_t)std::max(0, std::min(64 * MAX_LIST_REGS, atoi(t)));
	if (const char *t = getenv("VSS_FORCE_LOOPING"))
		h->force_looping = atoi(t) != 0;
	if (const char *t = getenv("VSS_EXACT_PROBE"))
		h->exact_probe = (uint32_t)atoi(t);
	if (const char *t = getenv("VSS_EXACT_KERNEL"))
		h->exact_kernel = (uint32_t)std::max(1, std::min(5, atoi(t)));
	if (const char *t = getenv("VSS_SEARCH_WGS_PER_CU"))
		h->search_wgs_per_cu = (uint32_t)std::max(1, std::min(8, atoi(t)));
	if (const char *t = getenv("VSS_SEARCH_WALKERS_CAP"))
		h->search_walkers_cap = (uint32_t)std::max(1, std::min((int)ENGINE_MAX_WALKERS, atoi(t)));
	if (const char *t = getenv("VSS_EXACT_FILTER"))
		h->exact_filter = atoi(t) != 0;
	if (const char *t = getenv("VSS_SEARCH_CREW_TUNE"))
		h->search_crew_tune = (uint32_t)atoi(t) & (CREW_SPARE_SIMD | CREW_NO_REQUESTS);
	if (const char *t = getenv("VSS_SEARCH_TOUCH_LISTS"))
		h->search_touch_lists = atoi(t) != 0;
	if (const char *t = getenv("VSS_SEARCH_TOUCH_ROWS"))
		h->search_touch_rows = atoi(t) != 0;
	if (const char *t = getenv("VSS_SEARCH_TOUCH_MAX"))
		h->search_touch_max_queries = (uint32_t)std::max(0, atoi(t));
	*out = h;
	return VSS_OK;
}

void vss_destroy(vss_index *h) {
	if (!h)
		return;
	(void)hipSetDevice(h->device);
	{
		// nobody may be waiting on (or name) this index's contexts any more (an index launches on its own device only)
		DeviceGate &dg = g_gate[(unsigned)h->device % 64];
		std::lock_guard<std::mutex> lk(dg.mu);
		if (dg.last.owner == h)
			dg.last = GateEntry();
	}
	if (h->stream)
		(void)hipStreamSynchronize(h->stream);
	h->release();
	if (h->own_stream && h->stream)
		(void)hipStreamDestroy(h->stream);
	delete h;
}

const char *vss_last_error(vss_index *h) {
	return h ? tls_error.c_str() : "null index";
}

int vss_set_stream(vss_index *h, void *s) {
	VSS_GUARD(h, {
		HIP_TRY(hipStreamSynchronize(h->stream));
		if (s) {
			if (h->own_stream)
				(void)hipStreamDestroy(h->stream);
			h->stream = (hipStream_t)s;
			h->own_stream = false;
		} else if (!h->own_stream) {
			HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
			h->own_stream = true;
		}
		return VSS_OK;
	})
}

int vss_synchronize(vss_index *h) {
	VSS_GUARD(h, {
		HIP_TRY(hipStreamSynchronize(h->stream));
		return VSS_OK;
	})
}

int vss_reserve(vss_index *h, uint64_t members, uint64_t threads) {
	VSS_GUARD(h, { return h->reserve(members, threads); })
}

int vss_stage_batch(vss_index *h, const int64_t *rowids, const float *vecs, const uint64_t *validity, uint64_t n) {
	VSS_GUARD(h, { return h->stage(rowids, vecs, validity, n, false); })
}

int vss_stage_batch_device(vss_index *h, const int64_t *rowids, const float *vecs, uint64_t n) {
	VSS_GUARD(h, { return h->stage(rowids, vecs, nullptr, n, true); })
}

int vss_build_finalize(vss_index *h) {
	VSS_GUARD(h, { return h->build_finalize(); })
}

int vss_add_batch(vss_index *h, const int64_t *rowids, const float *vecs, const uint64_t *validity, uint64_t n) {
	VSS_GUARD(h, {
		int rc = h->stage(rowids, vecs, validity, n, false);
		if (rc != VSS_OK)
			return rc;
		return h->build_finalize();
	})
}

int vss_set_build_reorder(vss_index *h, int on) {
	VSS_GUARD(h, {
		h->reorder_after_build = on != 0;
		return VSS_OK;
	})
}

int vss_set_build_params(vss_index *h, uint64_t max_batch, uint64_t growth_div) {
	VSS_GUARD(h, {
		if (!max_batch || !growth_div)
			return h->fail("max_batch and growth_div must be positive");
		h->max_batch = max_batch;
		h->growth_div = growth_div;
		return VSS_OK;
	})
}

#ifdef VSS_PARANOID
/* debug builds only: the 64 pinned words the kernels leave notes in (readable while a kernel runs) */
uint32_t *vss_debug_buffer(vss_index *h) {
	return h ? h->h_debug : nullptr;
}
#endif

int vss_set_option(vss_index *h, const char *name, int64_t value) {
	VSS_GUARD(h, {
		if (!name)
			return h->fail("vss_set_option: no option name");
		if (const char *why = set_option_checked(h, name, value))
			return h->fail("vss_set_option(\"%s\", %lld): %s", name, (long long)value, why);
		return VSS_OK;
	})
}

int vss_search(vss_index *h, const float *q, uint64_t k, uint64_t ef, int64_t *out, uint64_t *out_count) {
	VSS_SHARED(h, {
		uint32_t cnt = 0;
		int rc = h->search_host(q, 1, k, ef, out, nullptr, &cnt, false);
		if (out_count)
			*out_count = cnt;
		return rc;
	})
}

int vss_search_batch(vss_index *h, const float *Q, uint64_t nq, uint64_t k, uint64_t ef, int64_t *out, float *out_d,
                     uint32_t *out_counts) {
	VSS_SHARED(h, { return h->search_host(Q, nq, k, ef, out, out_d, out_counts, false); })
}

int vss_search_batch_device(vss_index *h, const float *Q, uint64_t nq, uint64_t k, uint64_t ef, int64_t *out,
                            float *out_d, uint32_t *out_counts) {
	VSS_SHARED(h, { return h->search_launch(0, Q, (uint32_t)h->dim, nq, k, ef, out, out_d, out_counts, false); })
}

int vss_search_batch_filtered(vss_index *h, const float *Q, uint64_t nq, uint64_t k, uint64_t ef, const uint64_t *allowed,
                              uint64_t n_bits, int64_t *out, float *out_d, uint32_t *out_counts) {
	VSS_SHARED(h, {
		if (!allowed)
			return h->fail("filtered search needs a row-id bitmap");
		return h->search_host(Q, nq, k, ef, out, out_d, out_counts, false, allowed, n_bits);
	})
}

int vss_search_batch_filtered_device(vss_index *h, const float *Q, uint64_t nq, uint64_t k, uint64_t ef,
                                     const uint64_t *d_allowed, uint64_t n_bits, int64_t *out, float *out_d,
                                     uint32_t *out_counts) {
	VSS_SHARED(h, {
		if (!d_allowed)
			return h->fail("filtered search needs a row-id bitmap");
		return h->search_launch(0, Q, (uint32_t)h->dim, nq, k, ef, out, out_d, out_counts, false, d_allowed, n_bits);
	})
}

// the caller's contexts are 0 .. EXPLICIT_CTX-1; the others are leased internally by the host-pointer entry points
#define VSS_CHECK_CALLER_CONTEXT(h, context)                                                                           \
	if ((context) < 0 || (context) >= vss_index::EXPLICIT_CTX)                                                         \
		return (h)->fail("search context %d out of range (0..%d)", (int)(context), vss_index::EXPLICIT_CTX - 1);

int vss_search_batch_device_begin(vss_index *h, int context, const float *Q, uint64_t nq, uint64_t k, uint64_t ef,
                                  int64_t *out, float *out_d, uint32_t *out_counts) {
	VSS_SHARED(h, {
		VSS_CHECK_CALLER_CONTEXT(h, context)
		if (nq && k && (!Q || !out || !out_counts))
			return h->fail("vss_search_batch_device_begin: null query / row-id / count pointer");
		return h->search_begin_multi(context, 1, &Q, (uint32_t)h->dim, nq, k, ef, &out, &out_d, &out_counts, nullptr, 0, false,
		                             true);
	})
}

int vss_search_multi_device_begin(vss_index *h, int context, uint64_t n_batches, const float *const *Q, uint64_t per_batch,
                                  uint64_t k, uint64_t ef, int64_t *const *out, float *const *out_d,
                                  uint32_t *const *out_counts) {
	VSS_SHARED(h, {
		VSS_CHECK_CALLER_CONTEXT(h, context)
		if (!Q || !out || !out_d || !out_counts)
			return h->fail("vss_search_multi_device_begin: null table");
		if (n_batches < 1 || n_batches > MAX_COALESCED)
			return h->fail("1 to %d batches per launch", MAX_COALESCED);
		for (uint64_t b = 0; per_batch && k && b != n_batches; ++b) // only d_out_distances[b] may be NULL
			if (!Q[b] || !out[b] || !out_counts[b])
				return h->fail("vss_search_multi_device_begin: null query / row-id / count pointer for batch %llu",
				               (unsigned long long)b);
		return h->search_begin_multi(context, n_batches, Q, (uint32_t)h->dim, per_batch, k, ef, out, out_d, out_counts, nullptr,
		                             0, false, true);
	})
}

int vss_search_batch_end(vss_index *h, int context) {
	VSS_SHARED(h, {
		VSS_CHECK_CALLER_CONTEXT(h, context)
		return h->search_end(context, false);
	})
}

int vss_search_exact_batch(vss_index *h, const float *Q, uint64_t nq, uint64_t k, int64_t *out, float *out_d,
                           uint32_t *out_counts) {
	VSS_SHARED(h, { return h->search_host(Q, nq, k, 0, out, out_d, out_counts, true); })
}

int vss_search_exact_batch_device(vss_index *h, const float *Q, uint64_t nq, uint64_t k, int64_t *out, float *out_d,
                                  uint32_t *out_counts) {
	VSS_SHARED(h, { return h->exact_launch(Q, (uint32_t)h->dim, nq, k, out, out_d, out_counts); })
}

int vss_last_search_stats(vss_index *h, uint64_t *out4) {
	VSS_SHARED(h, {
		std::lock_guard<std::mutex> lk(h->stats_mu);
		std::memcpy(out4, h->last_stats, sizeof h->last_stats);
		return VSS_OK;
	})
}

/* debug builds only: per-query shader-clock ticks of the last search (VSS_PHASE_STRIDE values per query) */
#ifdef VSS_PHASE_TIMERS
int vss_debug_phase_ticks(vss_index *h, unsigned long long *out, uint64_t nq) {
	VSS_SHARED(h, {
		if (!h->ctx[0].d_phase.p || h->ctx[0].d_phase.n < nq * VSS_PHASE_STRIDE)
			return h->fail("no phase ticks recorded for %llu queries", (unsigned long long)nq);
		HIP_TRY(hipMemcpy(out, h->ctx[0].d_phase.p, nq * VSS_PHASE_STRIDE * 8, hipMemcpyDeviceToHost));
		return VSS_OK;
	})
}
#endif

int vss_build_work(vss_index *h, uint64_t *out3) {
	VSS_SHARED(h, {
		std::memset(out3, 0, 3 * sizeof(uint64_t));
		if (h->d_work_stats.p) {
			HIP_TRY(hipStreamSynchronize(h->stream));
			unsigned long long v[4];
			HIP_TRY(hipMemcpy(v, h->d_work_stats.p, sizeof v, hipMemcpyDeviceToHost));
			out3[0] = v[0];
			out3[1] = v[1];
			out3[2] = v[2];
		}
		return VSS_OK;
	})
}

int vss_timing(vss_index *h, double *out6, int reset) {
	VSS_SHARED(h, { // timing[] is guarded by stats_mu (searches) and written by the build only under the exclusive lock
		std::lock_guard<std::mutex> lk(h->stats_mu);
		std::memcpy(out6, h->timing, sizeof h->timing);
		if (reset)
			std::memset(h->timing, 0, sizeof h->timing);
		return VSS_OK;
	})
}

int vss_last_search_query_stats(vss_index *h, uint32_t *out, uint64_t nq) {
	VSS_SHARED(h, {
		std::lock_guard<std::mutex> lk(h->stats_mu);
		if (h->last_query_stats.size() < nq * 2)
			return h->fail("no per-query stats recorded for %llu queries", (unsigned long long)nq);
		std::memcpy(out, h->last_query_stats.data(), nq * 8);
		return VSS_OK;
	})
}

int vss_remove_batch(vss_index *h, const int64_t *rowids, uint64_t n, uint64_t *removed) {
	VSS_GUARD(h, { return h->remove(rowids, n, removed); })
}

int vss_compact(vss_index *h) {
	VSS_GUARD(h, { return h->compact(true); })
}

int vss_compact_ex(vss_index *h, int reorder, int *out_reordered) {
	VSS_GUARD(h, {
		const int rc = h->compact(reorder != 0);
		if (out_reordered)
			*out_reordered = rc == VSS_OK && h->last_compact_reordered ? 1 : 0;
		return rc;
	})
}

uint64_t vss_size(vss_index *h) {
	// live rows: linked and not tombstoned, plus every staged row — appended (`staged`) or taking over a tombstoned slot
	// (`n_pending`; `tombstones` still counts that slot until vss_build_finalize has re-linked it)
	return h ? h->count + h->staged + h->n_pending - h->tombstones : 0;
}
uint64_t vss_nodes(vss_index *h) {
	return h ? h->count + h->staged : 0;
}
uint64_t vss_capacity(vss_index *h) {
	return h ? h->capacity : 0;
}
int vss_build_progress(vss_index *h, uint64_t *linked, uint64_t *total) {
	if (!h)
		return VSS_ERROR;
	if (linked)
		*linked = h->progress_linked.load(std::memory_order_relaxed);
	if (total)
		*total = h->progress_total.load(std::memory_order_relaxed);
	return VSS_OK;
}
uint64_t vss_max_level(vss_index *h) {
	return (h && h->count) ? (uint64_t)h->max_level : 0;
}
uint64_t vss_dimensions(vss_index *h) {
	return h ? h->dim : 0;
}
int vss_metric(vss_index *h) {
	return h ? h->metric : -1;
}
uint64_t vss_memory_usage(vss_index *h) {
	if (!h)
		return 0;
	return h->d_vectors.n * 4 + h->d_links0.n * 4 + h->d_links_up.n * 4 + h->d_upper_off.n * 4 + h->d_levels.n +
	       h->d_keys.n * 8 + h->d_list_owner.n * 4;
}

int vss_level_stats(vss_index *h, uint64_t level, uint64_t *out4) {
	VSS_SHARED(h, {
		HIP_TRY(hipStreamSynchronize(h->stream));
		h->level_stats(level, out4);
		return VSS_OK;
	})
}

uint64_t vss_serialized_length(vss_index *h) {
	return h ? h->serialized_length() : 0;
}

int vss_save(vss_index *h, vss_write_cb write, void *ctx) {
	// read-only: a checkpoint does not stall concurrent searches (two saves at once are fine too: each syncs the stream
	// for its own copies)
	VSS_SHARED(h, { return h->save(write, ctx); })
}

int vss_load(vss_index *h, vss_read_cb read, void *ctx) {
	VSS_GUARD(h, { return h->load(read, ctx); })
}

static int distance_launch(int fn, const float *a, const float *b, int b_const, uint64_t rows, uint64_t dim,
                           float *out, hipStream_t s) {
	if (fn < 0 || fn > 2 || !dim)
		return VSS_ERROR;
	if (!rows)
		return VSS_OK;
	const bool vec4 = (dim % 4 == 0) && ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0);
	const uint64_t units = vec4 ? dim / 4 : dim;
	const uint32_t G = (uint32_t)std::min<size_t>(64, ceil_pow2(units));
	const uint32_t logG = log2u(G);
	const uint64_t rows_per_block = 4 * (64 / G);
	const uint32_t grid = (uint32_t)std::min<uint64_t>((rows + rows_per_block - 1) / rows_per_block, 8192);
	if (vec4)
		hipLaunchKernelGGL(k_array_distance<true>, dim3(grid), dim3(256), 0, s, fn, a, b, b_const, rows, (uint32_t)dim,
		                   G, logG, out);
	else
		hipLaunchKernelGGL(k_array_distance<false>, dim3(grid), dim3(256), 0, s, fn, a, b, b_const, rows,
		                   (uint32_t)dim, G, logG, out);
	return hipGetLastError() == hipSuccess ? VSS_OK : VSS_ERROR;
}

int vss_distance_batch_device(int fn, const float *a, const float *b, int b_const, uint64_t rows, uint64_t dim,
                              float *out, void *stream) {
	return distance_launch(fn, a, b, b_const, rows, dim, out, (hipStream_t)stream);
}

// Host-pointer form: what a DuckDB scalar function sees is one <= 2048-row chunk of pageable host memory per call (SURVEY §8
// a13).  Round 6: the device buffers and the stream of a calling thread are kept between calls (grow-only, per thread and
// device) — rounds 1-5 paid three hipMalloc / hipFree pairs per chunk, which cost more than the chunk's transfer — and the
// operands travel as two asynchronous copies on that stream, the result as one.
namespace {
struct DistanceScratch {
	int device = -1;
	hipStream_t stream = nullptr;
	float *d_a = nullptr, *d_b = nullptr, *d_out = nullptr;
	uint64_t cap_a = 0, cap_b = 0, cap_out = 0; // floats
	void release() {
		if (device >= 0 && hipSetDevice(device) == hipSuccess) {
			(void)hipFree(d_a), (void)hipFree(d_b), (void)hipFree(d_out);
			if (stream)
				(void)hipStreamDestroy(stream);
		}
		d_a = d_b = d_out = nullptr, stream = nullptr, cap_a = cap_b = cap_out = 0, device = -1;
	}
	static bool grow(float *&p, uint64_t &cap, uint64_t want) {
		if (want <= cap)
			return true;
		(void)hipFree(p);
		p = nullptr, cap = 0;
		const uint64_t n = std::max<uint64_t>(want, 1u << 16);
		if (hipMalloc(&p, n * 4 + 16) != hipSuccess)
			return false;
		cap = n;
		return true;
	}
	~DistanceScratch() {
		release(); // (a thread that ends after the HIP runtime has been torn down gets errors back, nothing else)
	}
};
thread_local DistanceScratch t_distance;
} // namespace

int vss_distance_batch(int fn, const float *a, const float *b, int b_const, uint64_t rows, uint64_t dim, float *out,
                       int device) {
	if (hipSetDevice(device) != hipSuccess) {
		fprintf(stderr, "vssgpu: no HIP device %d — the engine has no CPU fallback\n", device);
		return VSS_ERROR;
	}
	if (fn < 0 || fn > 2 || !dim)
		return VSS_ERROR;
	if (!rows)
		return VSS_OK;
	DistanceScratch &sc = t_distance;
	if (sc.device != device) {
		sc.release();
		if (hipStreamCreateWithFlags(&sc.stream, hipStreamNonBlocking) != hipSuccess)
			return VSS_ERROR;
		sc.device = device;
	}
	const uint64_t nb = b_const ? dim : rows * dim;
	if (!DistanceScratch::grow(sc.d_a, sc.cap_a, rows * dim) || !DistanceScratch::grow(sc.d_b, sc.cap_b, nb) ||
	    !DistanceScratch::grow(sc.d_out, sc.cap_out, rows))
		return VSS_ERROR;
	int rc = VSS_ERROR;
	if (hipMemcpyAsync(sc.d_a, a, rows * dim * 4, hipMemcpyHostToDevice, sc.stream) == hipSuccess &&
	    hipMemcpyAsync(sc.d_b, b, nb * 4, hipMemcpyHostToDevice, sc.stream) == hipSuccess) {
		rc = distance_launch(fn, sc.d_a, sc.d_b, b_const, rows, dim, sc.d_out, sc.stream);
		if (rc == VSS_OK && hipMemcpyAsync(out, sc.d_out, rows * 4, hipMemcpyDeviceToHost, sc.stream) != hipSuccess)
			rc = VSS_ERROR;
	}
	if (hipStreamSynchronize(sc.stream) != hipSuccess)
		rc = VSS_ERROR;
	return rc;
}

// one 64-thread workgroup per query; the query's cells staged in LDS when they fit (exact_kernels.h: k_merge_topk)
static int merge_launch(const float *d, const int64_t *ids, size_t stride_d, size_t stride_id, uint64_t n_shards, uint64_t nq,
                        uint64_t k, float *out_d, int64_t *out_id, uint32_t *out_count, void *stream) {
	if (n_shards == 0 || n_shards * k > 0xFFFFFFFFull || nq > 0x7FFFFFFFull)
		return VSS_ERROR;
	const uint32_t head = ((uint32_t)n_shards * 4u + 15u) & ~15u;
	const uint64_t staged = head + n_shards * k * 12;
	if (staged <= vss::MERGE_STAGE_MAX_BYTES)
		hipLaunchKernelGGL(k_merge_topk<true>, dim3((uint32_t)nq), dim3(64), (uint32_t)staged, (hipStream_t)stream, d, ids, stride_d,
		                   stride_id, (uint32_t)n_shards, (uint32_t)nq, (uint32_t)k, out_d, out_id, out_count);
	else
		hipLaunchKernelGGL(k_merge_topk<false>, dim3((uint32_t)nq), dim3(64), head, (hipStream_t)stream, d, ids, stride_d, stride_id,
		                   (uint32_t)n_shards, (uint32_t)nq, (uint32_t)k, out_d, out_id, out_count);
	return hipGetLastError() == hipSuccess ? VSS_OK : VSS_ERROR;
}

int vss_merge_topk_device(const float *in_d, const int64_t *in_id, uint64_t n_shards, uint64_t nq, uint64_t k,
                          float *out_d, int64_t *out_id, uint32_t *out_count, void *stream) {
	if (!nq || !k)
		return VSS_OK;
	return merge_launch(in_d, in_id, (size_t)(nq * k), (size_t)(nq * k), n_shards, nq, k, out_d, out_id, out_count, stream);
}

uint64_t vss_packed_block_bytes(uint64_t nq, uint64_t k) {
	return (nq * k * 12 + 15) & ~15ull;
}

int vss_merge_topk_packed_device(const void *packed, uint64_t n_shards, uint64_t nq, uint64_t k, float *out_d,
                                 int64_t *out_id, uint32_t *out_count, void *stream) {
	if (!nq || !k)
		return VSS_OK;
	if ((uintptr_t)packed % 16)
		return VSS_ERROR;
	const uint64_t block = vss_packed_block_bytes(nq, k);
	const int64_t *ids = reinterpret_cast<const int64_t *>(packed);
	const float *d = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(packed) + nq * k * 8);
	return merge_launch(d, ids, (size_t)(block / 4), (size_t)(block / 8), n_shards, nq, k, out_d, out_id, out_count, stream);
}
}
