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
def oracle_lib():
    """Builds oracle/liboracle.so on first use (g++ only, a few seconds)."""
    from oracle import binding
    if not os.path.exists(binding.ORACLE_SO):
        binding.build_oracle()
    return binding


@pytest.fixture(scope="session")
def native_libs():
    """The product libraries must already be built in-tree (python -m raytracer_amd.build)."""
    from raytracer_amd import build
    missing = [p for p in (build.HIP_SO, build.HOST_SO) if not os.path.exists(p)]
    if missing:
        # CPU-side tests may build them (hipcc cross-compiles); GPU tests never build.
        build.build_all()
    return build
