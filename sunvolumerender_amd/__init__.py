"""MI355X-native drop-in for SunVolumeRender's render launch (path tracer + ray caster).

Layout: csrc/ = HIP kernels and the C-ABI (libsvr_hip.so); abi.py = ctypes binding of
include/svr_abi.h; host.py = the reference's host-side API (Canvas protocol, camera/volume
set-up) above the C ABI; scenes.py = deterministic synthetic scenes; dist.py = tile sharding
across GPUs with torch.distributed (RCCL).  There is no CPU rendering path in this package.
"""
from . import abi, host, scenes  # noqa: F401

__all__ = ["abi", "host", "scenes"]
__version__ = "0.1.0"
