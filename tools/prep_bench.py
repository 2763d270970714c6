#!/usr/bin/env python3
"""Time the volume-load preprocessing (svr_volume_preprocess: cast, range, gradient maximum, rescale, histogram)
on a synthetic 512^3 MET_SHORT volume resident in HBM, against the HBM roofline."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import abi, host, scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dtype = np.dtype(sys.argv[2]) if len(sys.argv) > 2 else np.dtype(np.int16)
vol = scenes.make_ct_head_volume(n)
hu = ((vol.astype(np.float32) / 65535.0) * 3000.0 - 1000.0).astype(dtype)
dev = host.Device(0)
src = dev.malloc(hu.nbytes)
dev.to_device(src, hu)
out = dev.malloc(hu.size * 2)
hist = np.zeros(65536, dtype=np.uint32)
info = abi.VolumeInfo()
sp = (C.c_double * 3)(1.0, 1.0, 1.0)
best = 1e9
for it in range(6):
    dev.check(dev.lib.svr_volume_preprocess(C.c_void_p(src), [np.dtype(t) for t in (np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.float32, np.float64)].index(dtype),
                                            n, n, n, sp, 1, C.c_void_p(out), (hist.ctypes.data_as(C.c_void_p) if '--nohist' not in sys.argv else None), 65536, C.byref(info)))
    ms, nbytes = C.c_float(0), C.c_uint64(0)
    dev.lib.svr_volume_preprocess_last_ms(C.byref(ms), C.byref(nbytes))
    best = min(best, ms.value)
print(f"svr_volume_preprocess {n}^3 {dtype.name}: {best:.3f} ms  {hu.size / best / 1e6:.1f} Gvoxel/s  "
      f"{nbytes.value / best / 1e6:.1f} GB/s algorithmic ({nbytes.value / hu.size:.0f} B/voxel) = {nbytes.value / best / 1e6 / 8000:.3f} of 8 TB/s; "
      f"range {info.range[0]:.0f}..{info.range[1]:.0f}, maxMagnitude {info.maxMagnitude:.0f}, bins {info.hist_bins}")
