"""Regenerates tests/golden/usearch_golden.npz from the REFERENCE's own usearch build
(oracle/_ref/libusearch_ref.so, compiled by oracle/Makefile from the headers under /root/reference).

Run in the authoring container only:   python tests/golden/make_golden.py
The .npz holds inputs' hashes and expected outputs (data, not source): stream hashes, level sequences,
result keys, f32 distance bit patterns and the reference's computed_distances / visited_members counters.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import golden_cases  # noqa: E402
from oracle_lib import load_ref  # noqa: E402

if __name__ == "__main__":
    ref = load_ref()
    if ref is None:
        sys.exit("reference build missing: run `make -C oracle ref` where /root/reference exists")
    res = golden_cases.run_all(ref)
    path = os.path.join(HERE, "usearch_golden.npz")
    np.savez_compressed(path, **res)
    print("wrote", path, os.path.getsize(path), "bytes,", len(res), "arrays")
