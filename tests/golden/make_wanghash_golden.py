#!/usr/bin/env python3
"""tests/golden/wanghash_ref.npz: inputs and outputs of the REFERENCE's own wangHash (pathtracer.cu:70-79), cut out of the
file where it lies at build time and compiled against the genuine cuda_runtime.h of the triton wheel
(oracle/ref_wanghash.cpp -> oracle/_ref/libref_wanghash.so).
Run in the container that has /root/reference:  python tests/golden/make_wanghash_golden.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import binding  # noqa: E402


def main():
    ref = binding.wanghash_ref()
    assert ref is not None, "needs /root/reference and a genuine cuda_runtime.h"
    rs = np.random.RandomState(78)
    # frame numbers are what render_pathtracer hashes (pathtracer.cu:302): 0 .. 4095, then powers of two and random words
    a = np.concatenate([np.arange(4096, dtype=np.uint64), 2 ** np.arange(32, dtype=np.uint64), 2 ** np.arange(1, 33, dtype=np.uint64) - 1,
                        rs.randint(0, 2 ** 32, 4096, dtype=np.uint64)]).astype(np.uint32)
    out = np.array([ref.ref_wang_hash(int(x)) for x in a], dtype=np.uint32)
    np.savez_compressed(ROOT / "tests" / "golden" / "wanghash_ref.npz", a=a, out=out)
    print(out[:4], len(out))


if __name__ == "__main__":
    main()
