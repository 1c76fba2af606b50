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


def test_a_run_below_the_agreement_bar_fails(capsys, tmp_path):
    side = str(tmp_path / "full.json")
    good = {"metric": "m", "cpu_baseline": {"agreement": {"queries": 8, "id_match_frac": 1.0, "rank_distance_max_rel_err": 1e-7}}}
    bench.finish(good, side)  # prints the line, returns
    assert json.loads(capsys.readouterr().out.strip().splitlines()[-1])["metric"] == "m"
    assert json.load(open(side)) == good  # the complete object: sidecar
    bad = {"metric": "m", "cpu_baseline": {"agreement": {"queries": 8, "id_match_frac": 0.9, "rank_distance_max_rel_err": 1e-7}}}
    with pytest.raises(SystemExit) as exc:
        bench.finish(bad, side)
    assert exc.value.code == 4
    bench.finish({"metric": "m", "cpu_baseline": None}, side)  # --no-cpu-baseline: nothing to gate on


def test_mismatching_cells_must_be_near_ties_in_the_reference_arithmetic():
    """Round 4: a cell whose row ids differ is explained only if the two rows' distances to the query, recomputed with ONE
    arithmetic (the baseline library's metric), differ by at most 1e-5 relative; anything else is an unexplained mismatch and
    fails the run."""
    rng = np.random.default_rng(11)
    dim, nq, k = 16, 6, 4
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    rows = {key: rng.standard_normal(dim).astype(np.float32) for key in range(100, 140)}
    rows[201] = rows[101].copy()                              # an exact twin of row 101: a tie the two sides may rank either way
    rows[202] = rows[102] + np.float32(0.25)                  # a different row, far from a tie

    def ref_distance(a, b):
        return float(np.sum((a.astype(np.float32) - b.astype(np.float32)) ** 2, dtype=np.float32))

    def fetch(keys):
        return {key: rows[key] for key in keys}

    keys = np.arange(100, 100 + nq * k).reshape(nq, k)
    d = np.array([[ref_distance(queries[i], rows[key]) for key in keys[i]] for i in range(nq)], dtype=np.float32)
    same = bench.agreement(keys, d, keys, d, "l2sq", 64, queries=queries, fetch_rows=fetch, ref_distance=ref_distance)
    assert same["mismatching_cells"] == 0 and same["unexplained_mismatches"] == 0 and bench.agreement_ok(same)
    tie = keys.copy()
    tie[0, 1] = 201                                            # the engine names the twin: explained
    a = bench.agreement(keys, d, tie, d, "l2sq", 64, queries=queries, fetch_rows=fetch, ref_distance=ref_distance)
    assert a["mismatching_cells"] == 1 and a["unexplained_mismatches"] == 0 and a["mismatch_max_rel_distance_gap"] == 0.0
    far = keys.copy()
    far[0, 2] = 202                                            # the engine names a row that is NOT a near-tie: unexplained
    b = bench.agreement(keys, d, far, d, "l2sq", 64, queries=queries, fetch_rows=fetch, ref_distance=ref_distance)
    assert b["unexplained_mismatches"] == 1 and b["unexplained_examples"][0]["engine_row"] == 202 and not bench.agreement_ok(b)
    short = keys.copy()
    short[3, 3] = -1                                           # the engine returned fewer rows than the reference: unexplained
    c = bench.agreement(keys, d, short, d, "l2sq", 64, queries=queries, fetch_rows=fetch, ref_distance=ref_distance)
    assert c["unexplained_mismatches"] == 1 and not bench.agreement_ok(c)
    # a cell beyond a near-tie can still be explained — by proof: the oracle in kernel mode (the traversal restated in the
    # engine's summation order) answers that query on the same graph and must return the engine's answer bit for bit
    same_as_engine = lambda idxs: ([far[i] for i in idxs], [d[i] for i in idxs])          # noqa: E731
    not_the_engine = lambda idxs: ([keys[i] for i in idxs], [d[i] for i in idxs])         # noqa: E731
    ok = bench.agreement(keys, d, far, d, "l2sq", 64, queries=queries, fetch_rows=fetch, ref_distance=ref_distance, replay=same_as_engine)
    assert ok["unexplained_mismatches"] == 0 and ok["cells_beyond_a_near_tie"] == 1 and ok["queries_replayed_in_wave_order"] == 1
    assert ok["queries_replay_identical_to_engine"] == 1
    no = bench.agreement(keys, d, far, d, "l2sq", 64, queries=queries, fetch_rows=fetch, ref_distance=ref_distance, replay=not_the_engine)
    assert no["unexplained_mismatches"] == 1 and no["queries_replay_identical_to_engine"] == 0
    bits = d.copy()
    bits[0, 0] = np.nextafter(bits[0, 0], np.float32(9))                                        # one ulp off: not the engine's answer
    ulp = lambda idxs: ([far[i] for i in idxs], [bits[i] for i in idxs])                       # noqa: E731
    assert bench.agreement(keys, d, far, d, "l2sq", 64, queries=queries, fetch_rows=fetch, ref_distance=ref_distance,
                           replay=ulp)["unexplained_mismatches"] == 1
    # the gate: an explained mismatch passes, an unexplained one fails the run (exit code 4) although 99.9 % of the cells agree
    many = np.tile(keys, (60, 1))
    md = np.tile(d, (60, 1))
    mq = np.tile(queries, (60, 1))
    fine = many.copy()
    fine[0, 1] = 201
    assert bench.agreement_ok(bench.agreement(many, md, fine, md, "l2sq", 64, queries=mq, fetch_rows=fetch, ref_distance=ref_distance))
    bad = many.copy()
    bad[0, 2] = 202
    e = bench.agreement(many, md, bad, md, "l2sq", 64, queries=mq, fetch_rows=fetch, ref_distance=ref_distance)
    assert e["id_match_frac"] > 0.99 and e["unexplained_mismatches"] == 1
    with pytest.raises(SystemExit) as exc:
        bench.finish({"metric": "m", "cpu_baseline": {"agreement": e}}, os.devnull)
    assert exc.value.code == 4


def test_operating_point_is_selected_with_a_margin_and_reported_on_other_queries():
    """select_ef: the smallest ef of the sweep whose selection recall clears the target by two standard errors."""
    def recalls_at_factory(means):
        def recalls_at(ef):  # 2048 per-query recalls of exactly this mean, standard error 0.05 / sqrt(2047) = 0.0011
            return (means[ef] + 0.05 * np.where(np.arange(2048) % 2, 1.0, -1.0)).tolist()
        return recalls_at
    means = {16: 0.60, 32: 0.90, 48: 0.9505, 64: 0.958, 96: 0.99}
    ef, mean, se, log = bench.select_ef(recalls_at_factory(means), sorted(means), 0.95)
    assert ef == 64 and mean - 2 * se >= 0.95 and [e["ef"] for e in log] == [16, 32, 48, 64]  # 0.9505 is not enough margin
    assert 0.001 < se < 0.0012
    # a fixed ef (--ef) is taken as it is, whatever its recall; a sweep that never gets there ends at its last entry
    assert bench.select_ef(recalls_at_factory(means), [32], 0.95)[0] == 32
    assert bench.select_ef(recalls_at_factory({16: 0.5, 32: 0.6}), [16, 32], 0.95)[0] == 32
    m, s = bench.mean_and_se([1.0, 0.5, 0.5, 1.0])
    assert m == 0.75 and s == pytest.approx(np.std([1, .5, .5, 1], ddof=1) / 2)


def test_rows_are_fetched_back_from_the_generator_by_key():
    """make_row_fetch regenerates the chunk a key lives in: the rows the exactness check compares are the rows that were
    staged (same seed, same chunk index, same chunk length)."""
    dev = torch.device("cpu")
    gen = bench.Mixture(3 * bench.CHUNK, 8, True, dev)
    fetch = bench.make_row_fetch(gen, 3 * bench.CHUNK, full_chunks=True)
    keys = [5, bench.CHUNK - 1, bench.CHUNK, 2 * bench.CHUNK + 17]
    got = fetch(keys)
    for key in keys:
        want = gen.rows(bench.DATA_SEED, key // bench.CHUNK, bench.CHUNK)[key % bench.CHUNK].numpy()
        assert np.array_equal(got[key], want) and got[key].dtype == np.float32
    # a last chunk staged at its own length (configs[1] at development sizes) is regenerated at that length
    short = bench.make_row_fetch(gen, bench.CHUNK + 1000, full_chunks=False)
    key = bench.CHUNK + 7
    assert np.array_equal(short([key])[key], gen.rows(bench.DATA_SEED, 1, 1000)[7].numpy())


def test_extra_configurations_are_well_formed():
    assert set(bench.EXTRA_CONFIGS) == {"c2", "c4", "c5", "a13", "reference_default_options", "build_efc256", "quality"}
    assert bench.DEFAULT_EXTRAS[-1] == "quality"  # the longest one goes last: the first to be skipped when the budget is short
    assert sorted(bench.DEFAULT_EXTRAS) == sorted(bench.EXTRA_CONFIGS) and bench.DEFAULT_EXTRAS[0] == "c5"
    for name, (argv, limit) in bench.EXTRA_CONFIGS.items():
        assert argv[0] == "--config" and argv[1] in (name, "c3") and 60 <= limit <= 900
        if argv[1] == "c3":  # an extra never starts extras of its own
            assert argv[argv.index("--extras") + 1] == "none"
    # the driver's window is 30 minutes; --extras-budget-s (22 minutes) stops starting extras long before that
    assert sum(limit for _, limit in bench.EXTRA_CONFIGS.values()) <= 2800


def worst_case_result():
    """A headline result object with every field at its widest: long strings, 17-digit floats, 8 ranks, every optional key."""
    f = 123456.78901234567
    agreement = {"queries": 2048, "ef_search": 512, "id_match_frac": 0.99946289062512345, "query_match_frac": 0.99707031251234,
                 "rank_distance_max_rel_err": 2.0496419308605525e-06, "rank_distance_max_rel_err_all_cells": 2.0496419308605525e-06,
                 "distance_error_relative_to": "max(|d|, |1-d|)", "mismatching_cells": 123456, "unexplained_mismatches": 0,
                 "bars": {"id_match_frac_min": 0.99}, "mismatch_check": "x" * 600, "unexplained_examples": [{"query": 1}] * 4}
    return {
        "metric": "queries/sec at recall@10, 10M×768 FLOAT top-10; index build rows/sec", "config_id": "c3", "value": f * 7.3,
        "unit": "queries/s", "n_gpus": 8, "steps": 100000, "warmup": 10000, "ms_per_step": 1.1331028974382207,
        "higher_is_better": True, "scaling": "strong", "multi_gpu_mode": "replicated", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "recall_at_10": 0.9571, "recall_at_10_se": 0.00218, "recall": {"heldout": {"mean": 0.95}, "rule": "r" * 200},
        "recall_measured_on": "8192 held-out queries vs the exact MFMA path", "ef_search": 512,
        "repeat": {"queries_per_s": [f * 7.1, f * 7.3, f * 7.4], "median_over_value": 1.0123456789},
        "repeat_detail": {"queries_per_s": [f * 7.1, f * 7.3, f * 7.4], "runs": [f * 7.3] * 5, "first_region": f * 7.3,
                          "median_over_value": 1.0123456789, "frac_per_launch": [0.80123456, 0.83123456], "what": "w" * 200},
        "plateau": {"engine": 0.9412, "at_ef_search": 1536, "reference": None, "note": "n" * 300},
        "ef_sweep": [{"ef": e, "recall": 0.123456789, "se": 0.0012345678} for e in bench.EF_SWEEP],
        "build_rows_per_s": f, "build_s": 58.47133065800881,
        "build": {"rows_per_s": f, "M": 32, "ef_construction": 384, "distances_per_row": 10523.123456789,
                  "link_repair_distances_per_row": 1234.123456789, "algorithmic_MB_per_row": 32.123456789,
                  "phase_a_frac_of_hbm": 0.66123456789, "whole_build_frac_of_hbm": 0.6123456789},
        "build_kernel_ms": {"phase_a": f, "phase_b": f, "batches": 664, "retries": 0},
        "build_roofline": {"achieved": 5289.042701485045, "distances_per_row": 10523.1, "link_repair_distances_per_row": 1234.5},
        "host_api": {"threads": 4, "queries_per_s": f, "one_thread_queries_per_s": f, "what": "w" * 300},
        "small_launches": {"single_query": {"us_per_call": 391.7790425475687, "reference_thread_us_per_call": 2113.5005848105884},
                           "join_chunk": {"queries": 204, "us_per_call": 804.655287148697, "reference_thread_us_per_call": 431154.1},
                           "kernel": "k" * 200, "ef_search": 512},
        "rccl_ranks": 8, "collective_backend": "nccl", "collectives_per_launch": 1.0, "collectives_timed": 12345,
        "rank_devices": [{"rank": r, "device": r, "name": "AMD Instinct MI355X", "pci": "0000:%02x:00" % (r * 16), "uuid": "u" * 36}
                         for r in range(8)],
        "rank_pci": ["0000:%02x:00" % (r * 16) for r in range(8)],
        "config": {"workload": "configs[3]: 10M rows FLOAT[768] l2sq top-10, batched 1024 queries, " + "w" * 300, "rows": 10000000,
                   "dim": 768, "index_metric": "cosine", "k": 10, "batch_queries": 1024, "M": 32, "M0": 64, "ef_construction": 384,
                   "ef_search": 512, "batches_per_launch": 16, "batches_per_launch_timed": 10.123456789, "launches_in_flight": 3,
                   "launches_gated": True, "shards": 8, "reordered_after_build": False, "shards_per_gpu": 1,
                   "parallelism": "shard8-on-1-gpu"},
        "roofline": {"bound": "hbm", "kernel": "k_search", "achieved": 5875.174987948718, "peak": 8000.0, "unit": "GB/s",
                     "frac": 0.7343968734935897, "traffic": 67093634800.123, "traffic_over_algorithmic": 0.98808123456,
                     "traffic_source": "profiles/" + "p" * 200, "algorithmic_bytes_per_launch": 67902563974.123,
                     "avg_kernel_ms": 11.557538986206055, "launches": 12345, "effective_gbs_over_wall": 5992.6211580181935,
                     "frac_over_wall": 0.7490776447522742,
                     "regimes": [{"batches_per_launch": g, "launches_in_flight": p, "gated": True, "ms_per_step": 1.5196885409144063,
                                  "queries_per_s": f, "avg_kernel_ms": 1.48, "gbs_per_launch": 4583.59, "frac_per_launch": 0.5729,
                                  "gbs_over_wall": 4466.57, "frac_over_wall": 0.5583} for g in (1, 4, 8, 16) for p in (1, 2, 3)],
                     "distances_per_query": 2148.44326171875, "expansions_per_query": 86.531884765625, "visited_set": "v" * 200},
        "cpu_baseline": {"value": 473.1486743778781, "unit": "queries/s", "cores": 1, "kind": "reference",
                         "window_rates": [466.1, 473.1, 467.4], "sample": "s" * 400, "index_rows": 10000000, "agreement": agreement,
                         "build_rows_per_s": 558.1260751822688, "host_cores_available": 256, "cpu_model": "AMD EPYC 9575F 64-Core " * 3,
                         "all_cores": {"threads": 256, "search_queries_per_s": 2838.09, "note": "n" * 300}},
    }


def worst_case_extra():
    ag = {"queries": 2048, "id_match_frac": 0.99946289062512345, "query_match_frac": 0.99707031251234,
          "rank_distance_max_rel_err": 2.0496419308605525e-06, "mismatching_cells": 123456, "unexplained_mismatches": 0, "more": "m" * 500}
    return {"metric": "m" * 200, "config_id": "c5", "value": 174872.59574248834, "unit": "queries/s", "recall_at_100": 0.9568,
            "ef_search": 480, "build_rows_per_s": 265829.41132247646, "exit_code": 0, "wall_s": 88.8,
            "crud": [{"after": "a" * 60, "recall_at_100": 0.9571, "queries_per_s": 158921.39552997064}] * 3,
            "config": {"workload": "w" * 300, "rows": 12500000, "dim": 1536, "index_metric": "cosine", "k": 100},
            "roofline": {"kernel": "k" * 100, "frac": 0.6354871847233756, "avg_kernel_ms": 93.59112345, "distances_per_query": 4700.123456,
                         "expansions_per_query": 557.16123456, "visited_set": "v" * 100, "us_per_expansion": 2.461424784923191},
            "chunk_2048_us": {"768": 412.123456, "1536": 733.123456}, "cpu_chunk_2048_us": {"768": 1634.123456, "1536": 3301.123456},
            "crossover_rows": {"array_distance/768": 2048, "array_distance/1536": 2048, "array_cosine_distance/768": 2048,
                               "array_cosine_distance/1536": 2048},
            "plateau": {"M": 16, "ef_construction": 128, "rows": 200000, "at_ef_search": 1024, "engine": 0.9712, "reference": 0.9698},
            "quality_compact": {"16/128": [[e, 0.9123, 0.9101] for e in bench.QUALITY_EFS],
                                "32/384": [[e, 0.9123, 0.9101] for e in bench.QUALITY_EFS]},
            "cpu_baseline": {"value": 47.95082525984583, "kind": "reference", "agreement": ag}}


def test_the_last_stdout_line_is_the_compact_headline(capsys, tmp_path):
    """The driver parses the LAST JSON line of a bounded (8 000-character) stdout tail: round 4's 24 kB line was cut off and the
    round went unmeasured.  Whatever the field widths, the last line stays under 4 000 characters and keeps the contract's keys,
    `roofline` and `cpu_baseline`; the arrays and the extras are on earlier, small lines; the complete object is the sidecar."""
    result = worst_case_result()
    assert len(json.dumps(result)) > 8000  # what round 4 printed as ONE line
    # (an extra is either measured or broken: the last one stands for a broken one)
    extras = [(name, dict(worst_case_extra(), config_id=name)) for name in bench.DEFAULT_EXTRAS[:-1]] + \
             [(bench.DEFAULT_EXTRAS[-1], {"error": "e" * 900})]
    side = str(tmp_path / "full.json")
    last = bench.emit(result, side, extras)
    lines = capsys.readouterr().out.splitlines()
    assert all(ln.startswith("{") for ln in lines) and json.loads(lines[-1]) == last
    assert len(lines[-1]) < 4000 and len(lines[-1]) <= bench.LINE_LIMIT
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "recall_at_10", "ef_search", "build_rows_per_s"):
        assert key in last, key
    assert last["metric"] == result["metric"] and last["steps"] == 100000 and last["n_gpus"] == 8
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in last["roofline"], key
    assert "regimes" not in last["roofline"] and "ef_sweep" not in last
    for key in ("value", "unit", "cores", "kind", "cpu_model", "agreement"):
        assert key in last["cpu_baseline"], key
    assert set(last["cpu_baseline"]["agreement"]) == set(bench.AGREEMENT_SCALARS)
    assert last["config"]["workload"].startswith("configs[3]") and last["config"]["rows"] == 10000000
    # every other line: a small object of its own that cannot be mistaken for a headline; all of them and the headline together
    # fit the driver's tail
    side_lines = [json.loads(ln) for ln in lines[:-1]]
    assert all(("detail" in ln) != ("extra" in ln) and "metric" not in ln for ln in side_lines)
    assert all(len(ln) <= bench.SIDE_LINE_LIMIT for ln in lines[:-1])
    assert {ln["extra"] for ln in side_lines if "extra" in ln} == set(bench.DEFAULT_EXTRAS)
    assert {"ef_sweep", "regime", "small_launches", "host_api", "build"} <= {ln.get("detail") for ln in side_lines}
    # the extras' lines come right before the headline: at their widest they and the headline still fit the driver's tail
    # (round 6: and the spread of the timed region, printed once more between them)
    assert json.loads(lines[-2])["detail"] == "repeat" and len(json.loads(lines[-2])["queries_per_s"]) == 3
    assert sum(len(ln) + 1 for ln in lines if '"extra"' in ln[:9]) + len(lines[-2]) + 1 + len(lines[-1]) < 8000
    assert "repeat" in last and last["repeat"]["queries_per_s"] == pytest.approx(result["repeat"]["queries_per_s"], rel=1e-5)
    assert json.load(open(side))["roofline"]["regimes"] == result["roofline"]["regimes"]  # nothing is lost: the sidecar has it all
    # rounding keeps six significant digits; nothing in the line is wider than that
    assert last["value"] == pytest.approx(result["value"], rel=1e-5) and len(repr(last["value"])) <= 9
    # a result with only the contract's keys (the smallest line) passes through unchanged
    tiny = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0}
    assert bench.compact_line(tiny) == tiny
    # a line limit tighter than the optional parts: they are shed, the contract's keys stay
    tight = bench.compact_line(result, limit=2200)
    assert len(json.dumps(tight)) <= 2200 and "roofline" in tight and "cpu_baseline" in tight and "sample" not in tight["cpu_baseline"]


def test_visited_set_form_follows_the_engines_sizing_rule(monkeypatch):
    """The JSON line says where the walkers' visited sets live at the line's limit."""
    monkeypatch.delenv("VSS_VISITED_COMPACT", raising=False)
    assert bench.visited_set_form(60, 10_000_000).startswith("32-bit cells in LDS (64")
    assert bench.visited_set_form(200, 10_000_000).startswith("32-bit cells in LDS (32")
    assert bench.visited_set_form(480, 12_500_000).startswith("16-bit cells")
    assert bench.visited_set_form(480, 20_000_000).startswith("16-bit cells")     # slots of 25 bits: the wider key form (round 6)
    assert bench.visited_set_form(480, 40_000_000) == "32-bit cells in HBM"      # slots beyond 25 bits
    assert bench.visited_set_form(600, 12_500_000) == "32-bit cells in HBM"      # the list itself lives in HBM there
    monkeypatch.setenv("VSS_VISITED_COMPACT", "0")
    assert bench.visited_set_form(480, 12_500_000) == "32-bit cells in HBM"
