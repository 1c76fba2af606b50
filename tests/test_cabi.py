"""CPU-side checks of the drop-in boundary (no GPU needed): the shared library loads, exports every symbol
include/vssgpu.h declares, fails loudly without a device, and never depends on oracle/."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    from __graft_entry__ import load_package
    p = load_package()
    p.build_library()
    return p


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vssgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vss_[a-z0-9_]+)\s*\(", text)) - {"vss_write_cb", "vss_read_cb"})


def test_header_symbols_are_exported_and_bound(pkg):
    lib = pkg.load_library()
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "declared in include/vssgpu.h but not exported: " + n
    assert set(pkg.SIGNATURES) == set(names), set(pkg.SIGNATURES) ^ set(names)
    assert b"gfx950" in lib.vss_version()


def test_library_is_self_contained(pkg):
    """The product library must not link or reference the test-only oracle."""
    out = subprocess.run(["ldd", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "usearch" not in out
    assert "rccl" not in out  # the exchange dlopen()s RCCL on first use: no link-time dependency (vss_exchange.hip)
    syms = subprocess.run(["nm", "-D", "--undefined-only", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert "orc_" not in syms
    # sources mention the oracle in comments only: no #include of, or call into, anything under oracle/
    for f in os.listdir(pkg.CSRC):
        for line in open(os.path.join(pkg.CSRC, f)):
            code = line.split("//")[0]
            assert "oracle" not in code and "orc_" not in code, (f, line)
    for f in ("__init__.py",):
        assert "oracle" not in open(os.path.join(pkg.HERE, f)).read()


def test_exchange_entry_points_answer_without_a_gpu(pkg):
    """vss_exchange_available() only tries to load an RCCL library (no device needed); the other entry points refuse bad
    arguments with a message instead of crashing."""
    lib = pkg.load_library()
    have = lib.vss_exchange_available()
    assert have in (0, 1)
    if not have:
        assert b"rccl" in lib.vss_exchange_last_error().lower()
    assert lib.vss_exchange_allgather(None, None, None, 0, None) != 0
    assert lib.vss_exchange_last_error()  # a message either way (no library, or bad arguments)
    assert lib.vss_exchange_ranks(None) == 0 and lib.vss_exchange_destroy(None) == 0
    out = (C.c_void_p * 2)()
    assert lib.vss_exchange_init_all(out, 2, (C.c_int * 2)(0, 0)) != 0  # one rank per device (or no library)


def test_fails_loudly_without_a_device(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.VssError, match="no CPU fallback"):
        pkg.GpuIndex(8, "l2sq")
    import numpy as np
    with pytest.raises(pkg.VssError):
        pkg.distance_batch("array_distance", np.zeros((2, 4), np.float32), np.zeros(4, np.float32))


def test_kernels_build_for_gfx950_only(pkg):
    """No compatibility layers: the device code embedded in the library targets gfx950 and nothing else."""
    blob = open(pkg.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_hot_search_kernels_spill_no_registers(pkg):
    """The search engine's instantiations that answer the measured configurations must not use scratch memory: a spilled
    register inside the accept loop of the 8-register list cost limits of 257-512 a twelfth of their rate in round 6
    (profiles/r06g_*), and nothing but the code object's metadata shows it.  (The 16-wave variant of the 8-register list — an
    A/B option since round 5 — is known to spill and is not on any default path.)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    seen = 0
    for name, vgpr, agpr, sgpr, scratch, lds in kernel_resources.resources(pkg.LIB_PATH):
        m = re.match(r"void vss::k_search<(\d+), (\d+), (\d+), (\d+), (\d+)>", name)
        if not m:
            continue
        E, threads = int(m.group(4)), int(m.group(5))
        if (E in (1, 2, 4) and threads == 1024) or (E == 8 and threads == 768):
            seen += 1
            assert scratch == 0, (name, vgpr, sgpr, scratch)
            assert vgpr <= (128 if threads == 1024 else 168), (name, vgpr)
    assert seen >= 50, seen
