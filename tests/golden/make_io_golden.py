#!/usr/bin/env python3
"""Generates tests/golden/io_golden.npz: .hdr files with the floats the REFERENCE's stb_image.h (v2.12,
utils/stb_image.h, built where it lies into oracle/_ref/libstb_ref.so) decodes from them, and RGBA images with the
TGA bytes the reference's stb_image_write.h (v1.02) writes for them.  Data only: inputs and expected outputs.
Run from the repo root in the container that has /root/reference:  python tests/golden/make_io_golden.py"""
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import binding                                   # noqa: E402
from tests.test_io_cpu import _hdr_cases, _tga_images        # noqa: E402


def main():
    assert binding.stb_ref() is not None, "needs /root/reference (oracle/_ref/libstb_ref.so)"
    out = {}
    with tempfile.TemporaryDirectory() as d:
        d = Path(d)
        for name, path in _hdr_cases(d).items():
            if name == "long_header":
                continue
            out[f"hdr_{name}_file"] = np.frombuffer(path.read_bytes(), dtype=np.uint8)
            out[f"hdr_{name}_rgb"] = binding.ref_loadf(path)
        for name, img in _tga_images().items():
            out[f"tga_{name}_img"] = img
            out[f"tga_{name}_file"] = np.frombuffer(binding.ref_write_tga(d / f"{name}.tga", img), dtype=np.uint8)
    np.savez_compressed(ROOT / "tests" / "golden" / "io_golden.npz", **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
