import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the C-ABI library and the oracle are build products (git-ignored): make sure they exist
    lib = os.path.join(ROOT, "schnorr-sig_amd", "csrc", "libschnorr_sig_amd.so")
    orc = os.path.join(ROOT, "oracle", "libschnorr_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def engine():
    import schnorr_sig_amd as ssa
    eng = ssa.Engine(0)
    yield eng
    eng.close()
