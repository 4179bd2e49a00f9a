"""pytest configuration: markers + shared checker fixtures.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, oracle vs the
reference build when oracle/_ref is present, host logic, C-ABI symbol checks.
`-m gpu` runs on the MI355X box: parity of the HIP path against the oracle,
always through the C ABI of libcxlspeckv.so.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.bindings import Oracle, build_oracle
    build_oracle()
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle.bindings import Reference, have_reference
    if not have_reference():
        pytest.skip("oracle/_ref/libspeckv_ref.so not built (needs /root/reference)")
    return Reference()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
