// visited_compact.h — the arithmetic of the compact visited set, shared by the device code (wave_primitives.h: VisitedSet) and
// the host (host_logic.h: when a launch takes it; tests/host_logic_probe.cpp: a sequential model of the set checked against
// std::set on the CPU).  No reference counterpart: usearch's growing_hash_set_gt (index.hpp:1018-1144) stores whole slots.
//
// 2^L cells of 16 bits.  A slot below 2^K (K = 24, or 25 for indexes of 2^24 .. 2^25 slots per GPU: round 6) goes through a
// permutation of the K-bit numbers (multiplication by an odd constant); the upper L bits of the image are the home cell, the
// lower T = K - L bits the tag; a cell stores (tag, d) with d the displacement from the home cell in D = 16 - T bits.  (tag, d)
// at cell c names home c - d and with it exactly one slot: no false positives; cells are never emptied, so a key is found by
// the probe sequence that placed it: no false negatives.  0xFFFF — displacement 2^D - 1, never stored — is "empty"; a key that
// would need it does not fit (the set then MOVES to a plain table: VisitedSet::migrate).
// The set's FORM is one word: L in bits 0-7, K in bits 8-15 (0 = 24, so that a plain cell count still means 24-bit keys).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define VSS_HD __host__ __device__ __forceinline__
#else
#define VSS_HD inline
#endif

namespace vss {
namespace compact_visited {

constexpr uint32_t KEY_BITS = 24, KEY_BITS_MAX = 25;
constexpr uint32_t ODD = 0x9E3779u; // 2^24 / golden ratio, odd: a permutation of the K-bit numbers for every K
constexpr uint32_t EMPTY16 = 0xFFFFu;
constexpr uint32_t MIN_CELLS_LOG2 = 9, MAX_CELLS_LOG2 = 16; // at least one displacement bit, at least eight tag bits

VSS_HD uint32_t make_form(uint32_t cells_log2, uint32_t key_bits) {
	return cells_log2 | (key_bits << 8);
}
VSS_HD uint32_t cells_log2_of(uint32_t form) {
	return form & 0xFFu;
}
VSS_HD uint32_t key_bits_of(uint32_t form) {
	return (form >> 8) ? (form >> 8) : KEY_BITS;
}
VSS_HD uint32_t tag_bits(uint32_t form) {
	return key_bits_of(form) - cells_log2_of(form);
}
// a form is usable when a cell keeps at least one displacement bit next to its tag
VSS_HD bool form_ok(uint32_t form) {
	const uint32_t L = cells_log2_of(form), K = key_bits_of(form);
	return L >= MIN_CELLS_LOG2 && L <= MAX_CELLS_LOG2 && K >= KEY_BITS && K <= KEY_BITS_MAX && K > L && K - L <= 15;
}
// the home cell of `key` and the content a cell must have to name it at displacement 0 (one step along the probe sequence
// = next cell, content + 1: the displacement sits in the low bits)
VSS_HD void home_of(uint32_t key, uint32_t form, uint32_t &cell, uint32_t &want) {
	const uint32_t T = tag_bits(form);
	const uint32_t image = (key * ODD) & ((1u << key_bits_of(form)) - 1);
	cell = image >> T;
	want = (image & ((1u << T) - 1)) << (16 - T);
}
VSS_HD bool placed_too_far(uint32_t want, uint32_t form) {
	const uint32_t dmask = (1u << (16 - tag_bits(form))) - 1;
	return (want & dmask) == dmask;
}
// The cells are invertible — (tag, displacement) at cell c names the home cell c - d, home and tag are the image, and the
// multiplication by an odd constant is a permutation — which is what lets a set that has outgrown its cells MOVE to a bigger
// table instead of being thrown away with the query's work (round 6: VisitedSet::migrate).
constexpr uint32_t odd_inverse(uint32_t a) { // a^-1 mod 2^32 by Newton's iteration (a odd)
	uint32_t x = a; // correct to 3 bits
	for (int i = 0; i < 5; ++i)
		x *= 2u - a * x;
	return x;
}
constexpr uint32_t ODD_INV = odd_inverse(ODD); // mod 2^32, hence mod 2^K for every K
static_assert(ODD * ODD_INV == 1u, "inverse of the key permutation");
// the key a NON-EMPTY cell holds: `content` = the 16 bits of cell `cell`
VSS_HD uint32_t key_of(uint32_t cell, uint32_t content, uint32_t form) {
	const uint32_t T = tag_bits(form), D = 16 - T;
	const uint32_t home = (cell - (content & ((1u << D) - 1))) & ((1u << cells_log2_of(form)) - 1);
	const uint32_t image = (home << T) | (content >> D);
	return (image * ODD_INV) & ((1u << key_bits_of(form)) - 1);
}

} // namespace compact_visited
} // namespace vss
