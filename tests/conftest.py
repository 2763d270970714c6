import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding

    binding.build()
    return binding.load()


@pytest.fixture(scope="session")
def hip_dev():
    """One HIP device context for the whole GPU test session (the C ABI keeps one scene per process)."""
    from sunvolumerender_amd import abi, host

    if not abi.library_path().exists():
        pytest.fail(f"{abi.library_path()} missing: the HIP extension must be built (no CPU fallback exists)")
    return host.Device(0, fatal_errors=False)
