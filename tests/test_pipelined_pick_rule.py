"""The pipelined level search's pick (hnsw_kernels.h: level_search_pipelined), restated on the CPU: the successor of an expansion
is told from its fresh scores BEFORE they are inserted —

    m    = the smallest fresh distance the radius test admits,   e = the best list entry still unexpanded
    next = the row of m if m < e.distance, else e

unless a tie decides the order by position (two fresh rows at m, a non-finite m, or m == e.distance): then the kernel accepts
first and picks the plain way.  Round 6 narrowed the tie test from "m equals ANY entry's distance" to "m equals e's distance":
an entry in front of e is expanded already, so a fresh row landing in front of its equal changes nothing about which entry is
the first unexpanded one.  Checked here against the plain order (accept every fresh row in lane order with the reference's
sorted_buffer insert — new before equal, the last entry falls off, index.hpp:880-891 — then take the first unexpanded entry) on
lists full of ties."""
import math
import random


def insert(lst, limit, d, row):
    """sorted_buffer_gt::insert: lower_bound position (before equal distances); rejected at `limit`; the last falls off"""
    pos = 0
    while pos < len(lst) and lst[pos][0] < d:
        pos += 1
    if pos >= limit:
        return
    lst.insert(pos, [d, row, False])
    del lst[limit:]


def accept(lst, limit, fresh):
    for d, row in fresh:  # lane order; each against the radius AT THAT MOMENT
        if len(lst) < limit or d < lst[-1][0]:
            insert(lst, limit, d, row)


def first_unexpanded(lst):
    for d, row, expanded in lst:
        if not expanded:
            return d, row
    return None


def kernel_pick(lst, limit, fresh):
    """returns ("tie", None) or ("next", row or None)"""
    radius = lst[-1][0] if lst else math.inf
    admitted = [(d, row) for d, row in fresh if len(lst) < limit or d < radius]
    e = first_unexpanded(lst)
    e_d = e[0] if e else math.inf
    if not admitted:
        return "next", (e[1] if e else None)
    m = min(d for d, _ in admitted)
    who = [row for d, row in admitted if not d > m]
    if len(who) > 1 or not math.isfinite(m) or m == e_d:
        return "tie", None
    return "next", (who[0] if m < e_d else e[1])


def test_pick_before_accept_equals_accept_then_pick_on_tie_ridden_lists():
    rng = random.Random(20260601)
    decided = ties_with_expanded = 0
    for trial in range(20000):
        limit = rng.choice([1, 2, 3, 5, 8, 16])
        span = rng.choice([3, 6, 12, 40])  # few distinct distances: ties everywhere
        lst = []
        for row in range(rng.randint(0, limit)):
            insert(lst, limit, float(rng.randint(0, span)), 1000 + row)
        # every entry in front of the first unexpanded one is expanded (how a search leaves its list)
        cut = rng.randint(0, len(lst))
        for i, ent in enumerate(lst):
            ent[2] = i < cut or rng.random() < 0.3
        fresh = [(float(rng.randint(0, span)), row) for row in range(rng.randint(0, 6))]
        how, nxt = kernel_pick(lst, limit, fresh)
        after = [list(x) for x in lst]
        accept(after, limit, fresh)
        plain = first_unexpanded(after)
        if how == "tie":
            continue  # the kernel itself takes the plain order
        decided += 1
        assert nxt == (plain[1] if plain else None), (trial, lst, fresh, nxt, plain)
        expanded_ds = {d for d, _, x in lst if x}
        adm = [d for d, _ in fresh if len(lst) < limit or d < lst[-1][0]]
        if adm and min(adm) in expanded_ds:
            ties_with_expanded += 1
    assert decided > 5000 and ties_with_expanded > 500, (decided, ties_with_expanded)
