import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle_lib import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def ref_lib():
    """The reference's own usearch build; only exists where /root/reference was present at build time."""
    from oracle_lib import load_ref
    lib = load_ref()
    if lib is None:
        pytest.skip("oracle/_ref/libusearch_ref.so not built (reference tree absent)")
    return lib
