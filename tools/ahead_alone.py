#!/usr/bin/env python3
"""How long does a 64-frame launch of frame-ahead tracing take when NOTHING runs beside it?  65 per-frame calls (the call for frame 64 starts
the batch for frames 127..190), then the host sleeps; read the duration of the last k_trace_tile launch from a rocprofv3 --kernel-trace run."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import abi, host, scenes  # noqa: E402
sc = scenes.make_scene(sys.argv[1] if len(sys.argv) > 1 else "c3")
dev = host.Device(0, fatal_errors=False)
c = host.Canvas(dev, sc.width, sc.height)
scenes.apply_to_canvas(sc, c)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    dev.set_option(getattr(abi, "OPT_" + k.upper()), int(v))
for rep in range(3):
    c.ReStartRender()
    for f in range(65):
        c.paint(sync=True)
    time.sleep(0.1)
    dev.lib.svr_get_kernel_time(None, None)
c.close()
