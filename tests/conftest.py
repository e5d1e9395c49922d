import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "prob-matrix-factorization_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Several tests use torch on the GPU (the hpf_pytorch model, caller-owned statistics tensors) in the
# same process as the engine.  torch bundles its own HIP runtime and must be mapped before
# libpmf_hip.so pulls in /opt/rocm's (pmf_hip.load() explains); the engine itself never imports torch
# -- tests/test_sharded_fit_gpu.py checks that on bench.py subprocesses.
try:
    import torch  # noqa: F401,E402
except Exception:  # pragma: no cover
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
