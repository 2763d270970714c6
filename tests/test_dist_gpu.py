"""Multi-process sharding over the PRODUCT path: two / three ranks, each a process of its own driving libsvr_hip.so on
GPU 0 through svr_set_row_shard (the box has one GPU; on the 8-GPU node every rank has its own), frame assembled on
rank 0 over a real process group (gloo here, RCCL in bench.py), tone-mapped with svr_hdr_to_ldr_frame.  The assembled
HDR frame (after 3 and again after 6 progressive frames) and the LDR image must be the oracle's, bit for bit."""
import numpy as np
import pytest

from tests.test_dist_cpu import reference_frames, run_ranks
from tests.util import assert_bit_exact

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,mode", [(2, "gather"), (3, "reduce")])
def test_hip_ranks_assemble_the_oracle_frame(tmp_path, world, mode):
    out = tmp_path / "assembled.npz"
    run_ranks(world, "hip", mode, out)
    got, ref = np.load(out), reference_frames()
    assert_bit_exact(got["hdr3"], ref["hdr3"], f"{world} HIP ranks, {mode}: assembled after 3 frames")
    assert_bit_exact(got["hdr6"], ref["hdr6"], f"{world} HIP ranks, {mode}: assembled again after 6 frames")
    assert np.array_equal(got["img6"], ref["img6"])


@pytest.mark.parametrize("world,strip", [(1, 16), (2, 8), (3, 16), (8, 16)])
def test_native_pack_unpack_assembles_the_frame(hip_dev, world, strip):
    """The device half of svr_assemble_frame without the wire (one process stands in for every rank; RCCL refuses two ranks on
    one GPU, and the box has one): each rank renders its strips (svr_set_row_shard), packs the rows it owns (svr_pack_strips),
    'root' unpacks every rank's rows (svr_unpack_strips) -- the assembled frame and its tone-mapped image are the single-GPU
    ones, bit for bit; without a communicator svr_assemble_frame reports an error instead of doing anything (world > 1)."""
    import ctypes as C

    from sunvolumerender_amd import abi, host, scenes
    from tests.util import oracle_frames

    dev = hip_dev
    sc = scenes.make_scene("tiny_head", trace_depth=2)
    W, H, N = sc.width, sc.height, 5
    ref_hdr, ref_img, _ = oracle_frames(sc, N)
    canvas = host.Canvas(dev, W, H)
    frame = dev.lib.svr_device_malloc(W * H * 12)
    packed = dev.lib.svr_device_malloc(W * H * 12)
    img = dev.lib.svr_device_malloc(W * H * 4)
    try:
        scenes.apply_to_canvas(sc, canvas)
        dev.check(dev.lib.svr_memset_device(C.c_void_p(frame), 0xFF, W * H * 12))          # every float of the frame must be written
        for rank in range(world):
            dev.check(dev.lib.svr_set_row_shard(strip, rank, world))
            canvas.ReStartRender()
            canvas.paint_frames(N, sync=True)
            n = dev.lib.svr_strip_rows_owned(H, strip, rank, world)
            assert n == len([y for y in range(H) if world == 1 or (y // strip) % world == rank])
            dev.check(dev.lib.svr_pack_strips(C.c_void_p(packed), C.c_void_p(canvas.renderParams.hdrBuffer), W, H, strip, rank, world))
            dev.check(dev.lib.svr_unpack_strips(C.c_void_p(frame), C.c_void_p(packed), W, H, strip, rank, world))
        dev.lib.svr_set_row_shard(0, 0, 1)
        dev.synchronize()
        got = dev.to_host(frame, (H, W, 3), np.float32)
        assert_bit_exact(got, ref_hdr, f"frame assembled from {world} ranks' packed strips")
        dev.check(dev.lib.svr_hdr_to_ldr_frame(C.c_void_p(img), C.c_void_p(frame), W, H))
        dev.synchronize()
        assert np.array_equal(dev.to_host(img, (H, W, 4), np.uint8), ref_img)
        # world == 1 needs no communicator: a plain copy
        if world == 1:
            dev.check(dev.lib.svr_assemble_frame(None, C.c_void_p(packed), C.c_void_p(frame), W, H, strip, 0, 1, 0))
            dev.synchronize()
            assert_bit_exact(dev.to_host(packed, (H, W, 3), np.float32), ref_hdr, "svr_assemble_frame, world 1")
        else:
            assert dev.lib.svr_assemble_frame(None, C.c_void_p(frame), C.c_void_p(canvas.renderParams.hdrBuffer), W, H, strip, 0, world, 0) != 0
            assert b"nccl_comm" in dev.lib.svr_last_error()
            dev.lib.svr_clear_error()
    finally:
        dev.lib.svr_set_row_shard(0, 0, 1)
        for ptr in (frame, packed, img):
            dev.lib.svr_device_free(C.c_void_p(ptr))
        canvas.close()
