"""CPU tests of bench.py's measurement helpers: the synthetic generator (any chunk can be regenerated from its seed — the
GPU box and the CPU baseline see the same rows), recall, and the reference-agreement gate of the `cpu_baseline` leg (a run
whose reference answers disagree with the engine's must FAIL, exit code 4)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_mixture_chunks_are_reproducible_and_independent():
    dev = torch.device("cpu")
    gen = bench.Mixture(10_000, 48, True, dev)
    a, b = gen.rows(bench.DATA_SEED, 3, 500), gen.rows(bench.DATA_SEED, 3, 500)
    assert torch.equal(a, b) and a.shape == (500, 48) and a.dtype == torch.float32 and a.is_contiguous()
    assert torch.allclose(a.norm(dim=1), torch.ones(500), atol=1e-5)  # cosine / ip configurations: unit rows
    assert not torch.equal(a, gen.rows(bench.DATA_SEED, 4, 500))      # another chunk
    assert not torch.equal(a, gen.rows(bench.QUERY_SEED, 3, 500))     # queries: disjoint seed
    again = bench.Mixture(10_000, 48, True, dev)                       # a second process regenerates the same centres
    assert torch.equal(again.rows(bench.DATA_SEED, 3, 500), a)
    raw = bench.Mixture(10_000, 48, False, dev).rows(bench.DATA_SEED, 0, 200)
    assert float((raw.norm(dim=1) - 1).abs().max()) > 1e-3            # l2sq configurations keep their norms
    # low intrinsic dimension (recall 0.95 has to be attainable): rows minus their centre live in a 32-dim subspace —
    # with two centres the 200 rows span at most 32 + 2 of the 48 dimensions
    two = bench.Mixture(4, 48, False, dev)
    x = two.rows(bench.DATA_SEED, 0, 200).double()
    assert two.k == 2 and int(torch.linalg.matrix_rank(x - x.mean(0), tol=1e-6)) <= bench.INTRINSIC_DIM + 2 < 48


def test_recall_at_k():
    truth = torch.tensor([[1, 2, 3, 4], [5, 6, 7, 8]])
    assert bench.recall_at_k(truth, truth) == 1.0
    assert bench.recall_at_k(torch.tensor([[4, 3, 2, 1], [5, 6, 0, 0]]), truth) == pytest.approx(0.75)


def test_reference_agreement_gate():
    rng = np.random.default_rng(3)
    keys = rng.integers(0, 10**6, size=(1000, 10))
    d = np.sort(rng.random((1000, 10)).astype(np.float32), axis=1)
    a = bench.agreement(keys, d, keys, d * np.float32(1 + 2e-6), "l2sq", 64)
    assert a["queries"] == 1000 and a["id_match_frac"] == 1.0 and a["query_match_frac"] == 1.0
    assert 1e-6 < a["rank_distance_max_rel_err"] < 3e-6 and bench.agreement_ok(a)
    # 2 % of the cells name another row (near-ties swapped): still above the 0.99 bar? no — 0.98 fails
    other = keys.copy()
    flip = rng.random(keys.shape) < 0.02
    other[flip] += 1
    b = bench.agreement(keys, d, other, d, "l2sq", 64)
    assert 0.97 < b["id_match_frac"] < 0.99 and not bench.agreement_ok(b)
    # ids equal but a distance off by 1e-4 relative: fails the 1e-5 bar
    far = d.copy()
    far[5, 3] *= np.float32(1 + 1e-4)
    c = bench.agreement(keys, d, keys, far, "l2sq", 64)
    assert c["id_match_frac"] == 1.0 and c["rank_distance_max_rel_err"] > 5e-5 and not bench.agreement_ok(c)
    # cosine / ip: d = 1 - s, so a tiny d is measured relative to max(|d|, |1 - d|), not to itself
    small = np.full((4, 10), 1e-7, dtype=np.float32)
    e = bench.agreement(keys[:4], small, keys[:4], small * np.float32(3), "cosine", 64)
    assert e["rank_distance_max_rel_err"] < 1e-6 and bench.agreement_ok(e)
    assert not bench.agreement_ok(None) and not bench.agreement_ok(bench.agreement(keys[:0], d[:0], keys[:0], d[:0], "l2sq", 64))


def test_a_run_below_the_agreement_bar_fails(capsys):
    good = {"metric": "m", "cpu_baseline": {"agreement": {"queries": 8, "id_match_frac": 1.0, "rank_distance_max_rel_err": 1e-7}}}
    bench.finish(good)  # prints the line, returns
    assert json.loads(capsys.readouterr().out.strip())["metric"] == "m"
    bad = {"metric": "m", "cpu_baseline": {"agreement": {"queries": 8, "id_match_frac": 0.9, "rank_distance_max_rel_err": 1e-7}}}
    with pytest.raises(SystemExit) as exc:
        bench.finish(bad)
    assert exc.value.code == 4
    bench.finish({"metric": "m", "cpu_baseline": None})  # --no-cpu-baseline: nothing to gate on
