#!/usr/bin/env python3
"""tests/golden/fresnel_ref.npz: inputs and outputs of the REFERENCE's own schlick_fresnel (core/bsdf/fresnel.h:10-15),
compiled where it lies against the genuine cuda_runtime.h of the triton wheel (oracle/ref_fresnel.cpp ->
oracle/_ref/libref_fresnel.so).  The only function of the render path that builds here without stand-ins.
Run in the container that has /root/reference:  python tests/golden/make_fresnel_golden.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import binding  # noqa: E402


def main():
    ref = binding.fresnel_ref()
    assert ref is not None, "needs /root/reference and a genuine cuda_runtime.h"
    rs = np.random.RandomState(77)
    ni = np.concatenate([np.full(256, 1.0), rs.uniform(0.5, 3.0, 256)]).astype(np.float32)
    no = np.concatenate([np.full(256, 2.5), rs.uniform(0.5, 3.0, 256)]).astype(np.float32)       # pathtracer.cu:30 IOR 2.5
    c = np.concatenate([np.linspace(-1, 1, 256), rs.uniform(-1, 1, 256)]).astype(np.float32)
    out = np.array([ref.ref_schlick_fresnel(float(a), float(b), float(x)) for a, b, x in zip(ni, no, c)], dtype=np.float32)
    np.savez_compressed(ROOT / "tests" / "golden" / "fresnel_ref.npz", ni=ni, no=no, cosin=c, out=out)
    print(out[:4], len(out))


if __name__ == "__main__":
    main()
