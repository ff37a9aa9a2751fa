import os
import sys

import pytest
import torch  # noqa: F401  -- before libptgpu is dlopen'ed: torch bundles its own HIP runtime, one runtime per process

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure): oracle/oracle.py over oracle/libptoracle.so."""
    from oracle import oracle
    oracle.build()
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu():
    """One libptgpu context on device 0 — no fallback: fails if the library or the GPU is missing."""
    from path_tracer_amd.gpu import GpuContext
    ctx = GpuContext(0)
    yield ctx
    ctx.close()
