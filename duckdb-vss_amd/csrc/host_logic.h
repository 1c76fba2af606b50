// host_logic.h — the pure host-side logic of the engine: no HIP, no device state, so that it can be compiled and tested
// on a CPU (tests/host_logic_probe.cpp, tests/test_host_logic.py) against the oracle and the reference build.
//
//   LevelRng        the reference's level generator (libstdc++ minstd_rand0 + generate_canonical<double,53>)
//   FreeRing        usearch's ring_gt as index_dense uses it for freed slots, wrap quirk included
//   KeyMap          rowid -> slot (lazy; deletes and duplicate checks)
//   batch_schedule  how a bulk build is cut into batch-synchronous steps
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace vss {
namespace host {

constexpr int64_t FREE_KEY_HOST = 0x7FFFFFFFFFFFFFFFll; // = VSS_FREE_KEY (include/vssgpu.h), usearch's free_key_
constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;               // = vss::EMPTY_SLOT

inline size_t ceil_pow2(size_t v) {
	size_t p = 1;
	while (p < v)
		p <<= 1;
	return p;
}
inline uint32_t log2u(size_t v) {
	uint32_t l = 0;
	while ((size_t(1) << l) < v)
		l++;
	return l;
}

// Level generator: libstdc++'s std::default_random_engine + uniform_real_distribution<double>, as used by
// usearch choose_random_level_ (index.hpp:3723-3727).  One stream per index (the reference keeps one per thread
// context, all identically seeded — SURVEY A.2); restarted by every growing reserve.
struct LevelRng {
	uint64_t x = 1;
	uint32_t next() {
		x = (x * 16807ull) % 2147483647ull;
		return (uint32_t)x;
	}
	double canonical() {
		const long double R = 2147483646.0L;
		double sum = 0, tmp = 1;
		for (int k = 0; k != 2; ++k) {
			sum += double(next() - 1u) * tmp;
			tmp = (double)((long double)tmp * R);
		}
		double ret = sum / tmp;
		if (ret >= 1.0)
			ret = std::nextafter(1.0, 0.0);
		return ret;
	}
	int level(double inv_log_m) {
		double r = -std::log(canonical()) * inv_log_m;
		return (int)(int16_t)r;
	}
	// the level as the engine stores it (u8)
	uint8_t stored_level(double inv_log_m) {
		const int lv = level(inv_log_m);
		return (uint8_t)(lv < 0 ? 0 : lv > 255 ? 255 : lv);
	}
};
inline double inverse_log_connectivity(uint64_t M) { // usearch index.hpp:3549
	return 1.0 / std::log((double)M);
}

// rowid -> slot, open addressing; built lazily (only deletes and duplicate checks need it)
struct KeyMap {
	std::vector<int64_t> k;
	std::vector<uint32_t> v;
	size_t mask = 0, used = 0;
	bool ready = false;
	static uint64_t hash(int64_t key) {
		uint64_t z = (uint64_t)key + 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		return z ^ (z >> 31);
	}
	void init(size_t n) {
		size_t cap = ceil_pow2(std::max<size_t>(16, n * 2));
		k.assign(cap, FREE_KEY_HOST);
		v.assign(cap, 0);
		mask = cap - 1;
		used = 0;
		ready = true;
	}
	void grow() {
		std::vector<int64_t> ok;
		std::vector<uint32_t> ov;
		ok.swap(k);
		ov.swap(v);
		init(ok.size());
		for (size_t i = 0; i != ok.size(); ++i)
			if (ok[i] != FREE_KEY_HOST && ov[i] != NO_SLOT)
				put(ok[i], ov[i]);
	}
	void put(int64_t key, uint32_t slot) {
		if ((used + 1) * 2 > k.size())
			grow();
		size_t h = hash(key) & mask;
		while (k[h] != FREE_KEY_HOST && k[h] != key)
			h = (h + 1) & mask;
		if (k[h] == FREE_KEY_HOST)
			used++;
		k[h] = key;
		v[h] = slot;
	}
	bool find(int64_t key, uint32_t &slot) const {
		size_t h = hash(key) & mask;
		while (k[h] != FREE_KEY_HOST) {
			if (k[h] == key) {
				slot = v[h];
				return slot != NO_SLOT;
			}
			h = (h + 1) & mask;
		}
		return false;
	}
	void erase(int64_t key) { // keep the key as a probe-chain marker, drop the slot
		size_t h = hash(key) & mask;
		while (k[h] != FREE_KEY_HOST) {
			if (k[h] == key) {
				v[h] = NO_SLOT;
				return;
			}
			h = (h + 1) & mask;
		}
	}
};

// The free list of tombstoned slots.  Restates usearch's ring_gt (index.hpp:1150-1277) as index_dense uses it
// (free_keys_, index_dense.hpp:463) INCLUDING its size() == 0 when the ring is exactly full: the order in which removed
// slots are handed back to later inserts is part of the reference's observable behaviour (which slot a row lands in).
struct FreeRing {
	std::vector<uint32_t> el;
	size_t cap = 0, head = 0, tail = 0;
	bool empty = true;
	size_t size() const {
		if (empty)
			return 0;
		return head >= tail ? head - tail : cap - (tail - head);
	}
	bool try_pop(uint32_t &v) {
		if (empty)
			return false;
		v = el[tail];
		tail = (tail + 1) % cap;
		empty = head == tail;
		return true;
	}
	void push(uint32_t v) {
		el[head] = v;
		head = (head + 1) % cap;
		empty = false;
	}
	bool reserve(size_t n) {
		if (n < size())
			return false;
		if (n <= cap)
			return true;
		n = std::max<size_t>(ceil_pow2(n), 64);
		std::vector<uint32_t> grown(n);
		size_t i = 0;
		while (try_pop(grown[i]))
			i++;
		el.swap(grown);
		cap = n, head = i, tail = 0;
		empty = i == 0;
		return true;
	}
	void clear() {
		head = tail = 0;
		empty = true;
	}
};


// Batch schedule of the bulk build (mirrored by oracle/hnsw_oracle.cpp `schedule`): batch = clamp(nodes / growth_div, 1,
// max_batch); a row whose level exceeds the current top level runs alone and becomes the entry (index.hpp:2769-2772).
// solo_row: the row that re-links the current entry slot, if any.  Its lists are blank while it is being re-linked, so
// batch mates descending from the entry would find nothing but the entry: it runs alone, like a level promotion.
inline std::vector<uint64_t> batch_schedule(uint64_t existing, int cur_max_level, const uint8_t *lv, uint64_t n,
                                      uint64_t max_batch, uint64_t growth_div, uint64_t solo_row = ~0ull) {
	std::vector<uint64_t> sizes;
	uint64_t i = 0, cur = existing;
	int ml = cur_max_level;
	while (i < n) {
		uint64_t b = 1;
		if (cur != 0) {
			b = std::max<uint64_t>(1, std::min(max_batch, cur / growth_div));
			uint64_t take = 0;
			while (take < b && i + take < n) {
				if ((int)lv[i + take] > ml || i + take == solo_row) {
					if (take == 0)
						take = 1;
					break;
				}
				take++;
			}
			b = take;
		}
		for (uint64_t j = 0; j != b; ++j)
			ml = std::max<int>(ml, lv[i + j]);
		sizes.push_back(b);
		i += b;
		cur += b;
	}
	return sizes;
}


} // namespace host
} // namespace vss
