"""Helpers shared by the -m gpu parity tests and tools/gpu_diag.py."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import datagen  # noqa: E402
from oracle_lib import CpuIndex, load_oracle, parse_stream  # noqa: E402


def pkg():
    from __graft_entry__ import load_package
    return load_package()


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def make_data(n, dim, metric, seed, nq=64):
    norm = metric != "l2sq"
    X = datagen.mixture(n, dim, seed, normalize=norm)
    Q = datagen.mixture(nq, dim, seed + 1, n_clusters=max(2, int(np.sqrt(n))), normalize=norm)
    return X, Q


def oracle_index(dim, metric, M=16, M0=None, efc=128, efs=64):
    """The CPU mirror of the kernels: wave summation order + kernel candidate lists."""
    return CpuIndex(load_oracle(), dim, metric, M, M0, efc, efs, order=1, wave=1)


def gpu_index(dim, metric, M=16, M0=None, efc=128, efs=64):
    return pkg().GpuIndex(dim, metric, M, M0, efc, efs)


def first_graph_difference(blob_a, blob_b, ignore_counts=False):
    """Human-readable location of the first difference between two serialized graphs (or None).
    ignore_counts: skip the header's count_present / count_deleted, which usearch mis-reports once 64 slots were freed
    (ring_gt::size() == 0 when full)."""
    if blob_a == blob_b:
        return None
    a, b = parse_stream(blob_a), parse_stream(blob_b)
    if ignore_counts:
        ha, hb = bytearray(a["head"]), bytearray(b["head"])
        ha[17:33] = hb[17:33] = bytes(16)
        if ha != hb:
            return "stream header differs"
    elif a["head"] != b["head"]:
        return "stream header differs"
    if a["rows"] != b["rows"]:
        return "row count %d vs %d" % (a["rows"], b["rows"])
    if not np.array_equal(a["levels"], b["levels"]):
        i = int(np.nonzero(a["levels"] != b["levels"])[0][0])
        return "level of slot %d: %d vs %d" % (i, a["levels"][i], b["levels"][i])
    if (a["max_level"], a["entry"]) != (b["max_level"], b["entry"]):
        return "entry/max_level %s vs %s" % ((a["max_level"], a["entry"]), (b["max_level"], b["entry"]))
    if not np.array_equal(a["keys"], b["keys"]):
        i = int(np.nonzero(a["keys"] != b["keys"])[0][0])
        return "key of slot %d: %d vs %d" % (i, a["keys"][i], b["keys"][i])
    if a["vectors"] is not None and not np.array_equal(a["vectors"], b["vectors"]):
        return "vector payload differs"
    n_bad = 0
    first = None
    for s in range(a["rows"]):
        for l in range(len(a["adj"][s])):
            if not np.array_equal(a["adj"][s][l], b["adj"][s][l]):
                n_bad += 1
                if first is None:
                    first = "slot %d level %d: %s vs %s" % (s, l, a["adj"][s][l].tolist(), b["adj"][s][l].tolist())
    if not n_bad:
        return None if ignore_counts else "streams differ outside the parsed fields"
    return "%d lists differ; first: %s" % (n_bad, first)


def recall_at_k(got, truth):
    k = truth.shape[1]
    return float(np.mean([len(set(got[i].tolist()) & set(truth[i].tolist())) / k for i in range(len(truth))]))
