// host_logic_probe.cpp — C surface over duckdb-vss_amd/csrc/host_logic.h (the engine's pure host logic) for the CPU tests
// in tests/test_host_logic.py.  Built with plain g++: nothing here needs HIP or a GPU.
#include <cstring>

#include "../duckdb-vss_amd/csrc/host_logic.h"

using namespace vss::host;

extern "C" {

// the first n levels the engine draws for connectivity M after a reserve (stage_metadata)
void hl_draw_levels(uint64_t M, uint64_t n, uint8_t *out) {
	LevelRng rng;
	const double inv = inverse_log_connectivity(M);
	for (uint64_t i = 0; i != n; ++i)
		out[i] = rng.stored_level(inv);
}

uint64_t hl_schedule(uint64_t existing, int max_level, const uint8_t *levels, uint64_t n, uint64_t max_batch,
                     uint64_t growth_div, int64_t solo_row, uint64_t *out_sizes) {
	auto s = batch_schedule(existing, max_level, levels, n, max_batch, growth_div, solo_row < 0 ? ~0ull : (uint64_t)solo_row);
	for (size_t i = 0; i != s.size(); ++i)
		out_sizes[i] = s[i];
	return s.size();
}

// Replays a script on the free ring exactly as the engine drives it: op > 0 = remove() of slot op-1 (reserve(size()+1) +
// push), op == 0 = one add() asking for a slot (try_pop).  out[i] = the slot an add received, or -1 (append), or -2 for
// remove ops.
void hl_ring_script(const int64_t *ops, uint64_t n, int64_t *out) {
	FreeRing ring;
	for (uint64_t i = 0; i != n; ++i) {
		if (ops[i] > 0) {
			ring.reserve(ring.size() + 1);
			ring.push((uint32_t)(ops[i] - 1));
			out[i] = -2;
		} else {
			uint32_t s = NO_SLOT;
			out[i] = ring.try_pop(s) ? (int64_t)s : -1;
		}
	}
}

// KeyMap: put keys[i] -> i, erase the flagged ones, look everything up again; returns the number of mismatches
uint64_t hl_keymap_check(const int64_t *keys, const uint8_t *erase, uint64_t n) {
	KeyMap m;
	m.init(4);
	for (uint64_t i = 0; i != n; ++i)
		m.put(keys[i], (uint32_t)i);
	for (uint64_t i = 0; i != n; ++i)
		if (erase[i])
			m.erase(keys[i]);
	uint64_t bad = 0;
	for (uint64_t i = 0; i != n; ++i) {
		uint32_t s = NO_SLOT;
		const bool found = m.find(keys[i], s);
		bad += erase[i] ? found : (!found || s != (uint32_t)i);
	}
	uint32_t s;
	bad += m.find(-12345, s);
	return bad;
}

// choose_search_shape over the engine's default policy with the given overrides;
// out = {solo, team, touch_lines, wants_solo, crew, roomy}
void hl_search_shape(uint32_t n, uint64_t M0, uint64_t V, uint64_t G, uint32_t solo_lds_bytes, uint32_t solo_mode, int team,
                     int touch_rows, int touch_lists, uint32_t n_cus, int force_looping, int crew, uint32_t engine_walkers,
                     uint32_t *out) {
	SearchShapePolicy p;
	p.solo_mode = solo_mode, p.team = team != 0, p.touch_rows = touch_rows != 0, p.touch_lists = touch_lists != 0;
	p.n_cus = n_cus, p.force_looping = force_looping != 0;
	p.crew = crew != 0, p.engine_walkers = engine_walkers;
	const SearchShape s = choose_search_shape(p, n, M0, V, G, solo_lds_bytes);
	out[0] = s.solo, out[1] = s.team, out[2] = s.touch_lines;
	out[3] = wants_solo(p, n, M0, V, G);
	out[4] = s.crew, out[5] = s.roomy;
}

// visited-set sizing of a SEARCH with this limit (cells per limit entry chosen by search_cells_per_limit)
uint32_t hl_search_visited_log2(uint64_t limit, uint32_t bump, uint64_t M0, uint64_t list_cap_max, uint32_t max_log2) {
	return visited_set_log2(limit, bump, M0, list_cap_max, search_cells_per_limit(limit), max_log2);
}
// when a launch takes the compact visited set, and with how many cells (0 = the plain 32-bit set)
uint32_t hl_compact_cells_log2(int plain_fits_lds, int solo, int register_list, uint64_t limit, uint64_t nodes, int first_pass,
                               uint32_t lds_table_log2) {
	return compact_visited_cells_log2(plain_fits_lds != 0, solo != 0, register_list != 0, limit, nodes, first_pass != 0, lds_table_log2);
}
// A sequential model of the compact set over visited_compact.h's arithmetic (the device code runs the same probe sequence with
// a compare-and-swap per cell): out[i] = 1 if keys[i] was present, 0 if it was inserted now, 2 if it could not be placed
// (displacement beyond its bits).  Returns the number of cells in use.
uint64_t hl_compact_model(const uint32_t *keys, uint64_t n, uint32_t cells_log2, uint8_t *out) { // (cells_log2: the set's FORM)
	const uint32_t cells = 1u << vss::compact_visited::cells_log2_of(cells_log2);
	std::vector<uint16_t> table((size_t)cells, (uint16_t)vss::compact_visited::EMPTY16);
	const uint32_t mask = cells - 1;
	uint64_t used = 0;
	for (uint64_t i = 0; i != n; ++i) {
		uint32_t c, want;
		out[i] = 2;
		for (vss::compact_visited::home_of(keys[i], cells_log2, c, want); !vss::compact_visited::placed_too_far(want, cells_log2);
		     c = (c + 1) & mask, ++want) {
			if (table[c] == want) {
				out[i] = 1;
				break;
			}
			if (table[c] == vss::compact_visited::EMPTY16) {
				table[c] = (uint16_t)want;
				out[i] = 0;
				++used;
				break;
			}
		}
	}
	return used;
}
// (home cell, tag) of a key: distinct keys below 2^24 must never share both
void hl_compact_home(const uint32_t *keys, uint64_t n, uint32_t cells_log2, uint32_t *cells, uint32_t *tags) {
	for (uint64_t i = 0; i != n; ++i) {
		uint32_t c, want;
		vss::compact_visited::home_of(keys[i], cells_log2, c, want);
		cells[i] = c;
		tags[i] = want >> (16 - vss::compact_visited::tag_bits(cells_log2));
	}
}
// ... and of the build's insert search (always 64 cells per entry)
uint32_t hl_build_visited_log2(uint64_t limit, uint32_t bump, uint64_t M0, uint64_t list_cap_max, uint32_t max_log2) {
	return visited_set_log2(limit, bump, M0, list_cap_max, 64, max_log2);
}
}
