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
