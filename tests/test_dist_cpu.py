"""N>1 path on CPU: gloo ranks render their interleaved strips with the oracle (standing in for the GPU kernel, which
is bit-identical to it), the frame is assembled on rank 0 with sunvolumerender_amd.dist.FrameAssembler (strip gather,
or the reduce(SUM) of BASELINE.json's north star), and compared with the single-process render.  The protocol
(tests/dist_worker.py) assembles twice -- after 3 and after 6 progressive frames -- so an assembly that sums in place
into a rank's live accumulator would be caught.  tests/test_dist_gpu.py runs the same worker over libsvr_hip.so."""
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(world, renderer, mode, out_path, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(ROOT / "tests" / "dist_worker.py"), renderer, mode, str(out_path)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=str(ROOT))
    assert res.returncode == 0, f"ranks failed ({res.returncode}):\n{res.stdout[-2000:]}\n{res.stderr[-4000:]}"


def reference_frames():
    sys.path.insert(0, str(ROOT))
    from oracle import binding
    from sunvolumerender_amd import scenes
    from tests import dist_worker

    sc = scenes.make_scene(dist_worker.SCENE[0], trace_depth=dist_worker.SCENE[1])
    o = binding.OracleScene(sc)
    ref = o.new_hdr()
    out = {}
    for f in range(6):
        o.render_pathtracer(ref, f)
        if f == 2:
            out["hdr3"] = ref.copy()
    out["hdr6"] = ref
    out["img6"] = o.hdr_to_ldr(ref)
    return out


@pytest.mark.parametrize("world,mode", [(2, "gather"), (2, "reduce"), (3, "gather")])
def test_strip_sharding_assembles_the_single_process_frame(tmp_path, world, mode):
    from tests.util import assert_bit_exact

    out = tmp_path / "assembled.npz"
    run_ranks(world, "oracle", mode, out)
    got, ref = np.load(out), reference_frames()
    assert_bit_exact(got["hdr3"], ref["hdr3"], f"{world}-rank {mode}: frame assembled after 3 frames")
    assert_bit_exact(got["hdr6"], ref["hdr6"], f"{world}-rank {mode}: frame assembled again after 6 frames")
    assert np.array_equal(got["img6"], ref["img6"])


def test_frame_assembler_single_process():
    import torch

    from sunvolumerender_amd import dist

    H, W = 37, 5
    hdr = torch.arange(H * W * 3, dtype=torch.float32)
    asm = dist.FrameAssembler(H, W, 8, 0, 1)
    assert torch.equal(asm.assemble(hdr).reshape(-1), hdr)
    # row bookkeeping: the ranks' rows partition the frame, strips of 8 interleaved
    for world in (2, 3, 8):
        rows = [dist.owned_rows(H, 8, r, world) for r in range(world)]
        assert sorted(np.concatenate(rows).tolist()) == list(range(H))
        assert all(((r // 8) % world == q).all() for q, r in enumerate(rows))
        assert dist.owned_row_count(H, 8, 0, world) == len(rows[0])
    a = dist.FrameAssembler(2048, 2048, 16, 1, 8)
    assert a.bytes_sent_per_rank() == 256 * 2048 * 12        # 6 MiB per peer, SURVEY.md 8(e)


def test_native_strip_index_maths_matches_dist_py():
    """svr_strip_rows_owned / svr_strip_row_to_y (plain host code of libsvr_hip.so: what svr_pack_strips, svr_unpack_strips and
    svr_assemble_frame -- the C-ABI counterpart of dist.FrameAssembler for C++ hosts -- index with) against dist.owned_rows,
    for frame heights that are and are not multiples of the strip, every rank of worlds 1..9; no GPU needed."""
    import ctypes as C

    sys.path.insert(0, str(ROOT))
    from sunvolumerender_amd import abi, dist

    lib = C.CDLL(str(abi.library_path()))
    lib.svr_strip_rows_owned.restype = C.c_uint32
    lib.svr_strip_rows_owned.argtypes = [C.c_uint32] * 4
    lib.svr_strip_row_to_y.restype = C.c_uint32
    lib.svr_strip_row_to_y.argtypes = [C.c_uint32] * 5
    for H in (1, 7, 16, 80, 100, 1024, 1031):
        for strip in (8, 16, 24):
            for world in range(1, 10):
                seen = np.zeros(H, dtype=np.int32)
                for rank in range(world):
                    rows = dist.owned_rows(H, strip, rank, world)
                    assert lib.svr_strip_rows_owned(H, strip, rank, world) == len(rows), (H, strip, rank, world)
                    ys = [lib.svr_strip_row_to_y(p, H, strip, rank, world) for p in range(len(rows))]
                    assert ys == rows.tolist(), (H, strip, rank, world)
                    assert lib.svr_strip_row_to_y(len(rows), H, strip, rank, world) == 0xFFFFFFFF
                    seen[rows] += 1
                assert (seen == 1).all()                      # the ranks' rows partition the frame
