"""Helpers for the host-I/O tests: MetaImage / Radiance writers built from the format descriptions (not from the
product's parser), so the files the loaders are tested on come from independent code."""
from __future__ import annotations

import zlib
from pathlib import Path

import numpy as np

MET_NAMES = {np.dtype(np.int8): "MET_CHAR", np.dtype(np.uint8): "MET_UCHAR", np.dtype(np.int16): "MET_SHORT",
             np.dtype(np.uint16): "MET_USHORT", np.dtype(np.int32): "MET_INT", np.dtype(np.uint32): "MET_UINT",
             np.dtype(np.float32): "MET_FLOAT", np.dtype(np.float64): "MET_DOUBLE"}


def write_mhd(path: Path, vol: np.ndarray, spacing=(1.0, 1.0, 1.0), *, local=False, msb=False, compressed=False,
              header_pad=0, header_size_minus_one=False, extra_lines=(), spacing_key="ElementSpacing", slices=False) -> Path:
    """vol is [z][y][x].  local: .mha style (ElementDataFile = LOCAL); header_pad: bytes of junk before the raw data
    (HeaderSize); slices: ElementDataFile = LIST with one raw file per slice."""
    path = Path(path)
    a = np.ascontiguousarray(vol)
    nz, ny, nx = a.shape
    data = a.astype(a.dtype.newbyteorder(">" if msb else "<")).tobytes()
    lines = ["ObjectType = Image", "NDims = 3", "BinaryData = True", f"BinaryDataByteOrderMSB = {'True' if msb else 'False'}",
             f"CompressedData = {'True' if compressed else 'False'}"]
    if compressed:
        data = zlib.compress(data, 6)
        lines.append(f"CompressedDataSize = {len(data)}")
    lines += ["TransformMatrix = 1 0 0 0 1 0 0 0 1", "Offset = -12.5 3 7", "CenterOfRotation = 0 0 0", "AnatomicalOrientation = RAI",
              f"{spacing_key} = {spacing[0]!r} {spacing[1]!r} {spacing[2]!r}", f"DimSize = {nx} {ny} {nz}",
              f"ElementType = {MET_NAMES[a.dtype]}"]
    lines += list(extra_lines)
    if header_size_minus_one:
        lines.append("HeaderSize = -1")
    elif header_pad:
        lines.append(f"HeaderSize = {header_pad}")
    if slices:
        assert not compressed and not local
        lines.append("ElementDataFile = LIST")
        per = nx * ny * a.dtype.itemsize
        for z in range(nz):
            name = f"{path.stem}_{z:03d}.raw"
            (path.parent / name).write_bytes(b"J" * header_pad + data[z * per:(z + 1) * per])
            lines.append(name)
        path.write_bytes(("\n".join(lines) + "\n").encode())
        return path
    if local:
        lines.append("ElementDataFile = LOCAL")
        path.write_bytes(("\n".join(lines) + "\n").encode() + data)
    else:
        raw = path.with_suffix(".zraw" if compressed else ".raw")
        lines.append(f"ElementDataFile = {raw.name}")
        path.write_bytes(("\n".join(lines) + "\n").encode())
        junk = b"J" * (header_pad if not header_size_minus_one else 37)
        raw.write_bytes(junk + data)
    return path


def float_to_rgbe(rgb: np.ndarray) -> np.ndarray:
    """Greg Ward's float -> RGBE: mantissas scaled by 256 / 2^e of the largest component."""
    rgb = np.asarray(rgb, dtype=np.float32)
    m = rgb.max(axis=-1)
    out = np.zeros(rgb.shape[:-1] + (4,), dtype=np.uint8)
    nz = m > 1e-32
    mant, e = np.frexp(m[nz])
    scale = (mant * 256.0 / m[nz])[..., None]
    out[nz, :3] = np.clip(rgb[nz] * scale, 0, 255).astype(np.uint8)
    out[nz, 3] = (e + 128).astype(np.uint8)
    return out


def _rle_channel(vals: np.ndarray) -> bytes:
    out, i, n = bytearray(), 0, len(vals)
    while i < n:
        run = 1
        while i + run < n and run < 127 and vals[i + run] == vals[i]:
            run += 1
        if run >= 4:
            out += bytes([128 + run, int(vals[i])])
            i += run
            continue
        j = i
        while j < n and j - i < 128:
            r = 1
            while j + r < n and r < 4 and vals[j + r] == vals[j]:
                r += 1
            if r >= 4:
                break
            j += 1
        out += bytes([j - i]) + bytes(int(v) for v in vals[i:j])
        i = j
    return bytes(out)


def write_hdr(path: Path, rgbe: np.ndarray, *, rle=True, header_extra=("EXPOSURE=1.0",), magic="#?RADIANCE") -> Path:
    """rgbe is [h][w][4] uint8.  rle: new-style run-length scanlines (only legal for 8 <= w < 32768)."""
    h, w, _ = rgbe.shape
    head = "\n".join([magic, "FORMAT=32-bit_rle_rgbe", *header_extra, "", f"-Y {h} +X {w}", ""]).encode()
    body = bytearray()
    if rle and 8 <= w < 32768:
        for y in range(h):
            body += bytes([2, 2, w >> 8, w & 255])
            for k in range(4):
                body += _rle_channel(rgbe[y, :, k])
    else:
        body += rgbe.tobytes()
    Path(path).write_bytes(head + bytes(body))
    return Path(path)
