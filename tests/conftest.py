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


_HIP_DEV = []


@pytest.fixture(autouse=True)
def _reset_library_options():
    """The C ABI keeps one scene and one option set per process (like the reference's globals): leave every test
    with the defaults, whatever the previous one set."""
    yield
    if _HIP_DEV:
        from sunvolumerender_amd import abi

        dev = _HIP_DEV[0]
        dev.lib.svr_clear_error()
        dev.lib.svr_set_row_shard(0, 0, 1)
        dev.lib.svr_set_render_window(0, 0, -1, -1)
        for key, val in ((abi.OPT_ENV_ON_ESCAPE, 0), (abi.OPT_KERNEL, abi.KERNEL_AUTO), (abi.OPT_COUNT, 0), (abi.OPT_EMPTY_SKIP, 1),
                         (abi.OPT_RAY_SKIP, 1), (abi.OPT_PIPELINE, 1), (abi.OPT_FRAME_AHEAD, 1), (abi.OPT_SKIP_TONEMAP, 0),
                         (abi.OPT_FRAMES_PER_WAVE_LOG2, -1), (abi.OPT_RAYCAST_LANES_LOG2, 3), (abi.OPT_FAST_MATH, 0), (abi.OPT_QUEUE, 1),
                         (abi.OPT_FOLD, 1), (abi.OPT_BOUND_CULL, 1), (abi.OPT_PARK_END, 32), (abi.OPT_GROUP_FRAMES, 64), (abi.OPT_LOCAL_MAJORANT, 0), (abi.OPT_LIGHT_CULL, 1), (abi.OPT_LM_TUNE, 0), (abi.OPT_LM_SUBCELLS, 1), (abi.OPT_PARK_CHEAP, 16), (abi.OPT_PINHOLE_FAST, 1), (abi.OPT_POOL, 1), (abi.OPT_TRIPS, 1), (abi.OPT_NAN_GUARD, 0), (abi.OPT_MACRO_SHIFT_MIN, 0), (abi.OPT_SPLIT, 0), (abi.OPT_ENV_NEE, 0), (abi.OPT_FAST_BOUND, 1)):
            dev.lib.svr_set_option(key, val)
        dev.lib.svr_clear_error()


@pytest.fixture(scope="session")
def hip_dev():
    """One HIP device context for the whole GPU test session (the C ABI keeps one scene per process)."""
    from sunvolumerender_amd import abi, host

    if not abi.library_path().exists():
        pytest.fail(f"{abi.library_path()} missing: the HIP extension must be built (no CPU fallback exists)")
    dev = host.Device(0, fatal_errors=False)
    _HIP_DEV.append(dev)
    return dev
