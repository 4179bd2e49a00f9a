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


# Seeds of the samples the full-size tests drew this run (tests/test_gpu_full_size.py: sample_seed): printed at the end of
# EVERY run, passing or not, so that a run can be repeated with SPECKV_SAMPLE_SEED=<seed>.
SAMPLE_SEEDS = []


def pytest_terminal_summary(terminalreporter):
    if SAMPLE_SEEDS:
        terminalreporter.write_line("full-size sample seeds (SPECKV_SAMPLE_SEED=<seed> repeats one): "
                                    + ", ".join(f"{name}={seed}" for name, seed in SAMPLE_SEEDS))
