import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import cpu_checker
    return cpu_checker.restatement()


@pytest.fixture(scope="session")
def reference_lib():
    import cpu_checker
    ref = cpu_checker.reference()
    if ref is None:
        pytest.skip("oracle/_ref/libsa_ref.so not built (needs the reference checkout)")
    return ref
