import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import abi, host, scenes
sc = scenes.make_scene("c3")
dev = host.Device(0)
c = host.Canvas(dev, sc.width, sc.height)
scenes.apply_to_canvas(sc, c, 0)
for world in (1, 2, 4, 8):
    dev.check(dev.lib.svr_set_row_shard(16, 0 if world == 1 else 3 % world, world))
    c.ReStartRender(); c.paint_frames(32); dev.synchronize()
    dev.set_option(abi.OPT_TIMING, 1); dev.check(dev.lib.svr_reset_kernel_time())
    c.ReStartRender(); dev.synchronize(); t0 = time.perf_counter()
    for _ in range(8): c.paint_frames(32)
    dev.synchronize(); wall = (time.perf_counter() - t0) / 8 * 1e3
    k_ms, k_n = dev.kernel_time()
    dev.set_option(abi.OPT_TIMING, 0)
    print(f"world {world}: wall {wall:.3f} ms/step, trace kernel {k_ms / k_n:.3f} ms ({k_n} launches)")
dev.lib.svr_set_row_shard(0, 0, 1)
c.close()
