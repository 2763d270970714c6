#!/usr/bin/env python3
"""Generate the committed golden fixtures from the CPU oracle.

The reference ships no fixtures and cannot be built here, so these vectors pin the ORACLE (and through it
the numeric contract) against accidental change; they are what travels to the GPU box.  Re-run only
when the contract is changed on purpose:  python tests/golden/make_golden.py
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import binding  # noqa: E402
from sunvolumerender_amd import scenes  # noqa: E402
from tests.util import oracle_frames  # noqa: E402

OUT = Path(__file__).resolve().parent

RENDER_CASES = [("tiny", 1, 3), ("tiny_head", 4, 2), ("tiny_bone", 6, 1)]


def main():
    lib = binding.load()
    # ---- function-level known-answer vectors ----
    rs = np.random.RandomState(1234)
    x_log = np.concatenate([np.float32(1.0) - rs.rand(64).astype(np.float32), np.float32(10.0) ** rs.uniform(-30, 30, 32).astype(np.float32),
                            np.array([0.0, 1.0, 2.0 ** -126, 2.0 ** -140, np.inf], dtype=np.float32)])
    x_exp = np.concatenate([rs.uniform(-90, 89, 96).astype(np.float32), np.array([0.0, -86.6, -86.7, 88.72, 88.73, -np.inf], dtype=np.float32)])
    x_trig = np.concatenate([rs.uniform(0, 6.2832, 64).astype(np.float32), rs.uniform(-100, 100, 32).astype(np.float32),
                             np.array([0.0, np.pi / 4, np.pi / 2, np.pi, 2 * np.pi], dtype=np.float32)])
    x_acos = np.concatenate([rs.uniform(-1, 1, 64).astype(np.float32), np.array([-1, -0.5, 0, 0.5, 1], dtype=np.float32)])
    y_at, x_at = rs.uniform(-3, 3, 64).astype(np.float32), rs.uniform(-3, 3, 64).astype(np.float32)
    xp, yp = rs.rand(64).astype(np.float32), rs.uniform(0.1, 30, 64).astype(np.float32)
    f1 = lambda fn, xs: np.array([fn(float(v)) for v in xs], dtype=np.float32)
    kat = {
        "x_log": x_log, "y_log": f1(lib.svo_logf, x_log),
        "x_exp": x_exp, "y_exp": f1(lib.svo_expf, x_exp),
        "x_trig": x_trig, "y_sin": f1(lib.svo_sinf, x_trig), "y_cos": f1(lib.svo_cosf, x_trig),
        "x_acos": x_acos, "y_acos": f1(lib.svo_acosf, x_acos),
        "y_at": y_at, "x_at": x_at, "r_atan2": np.array([lib.svo_atan2f(float(a), float(b)) for a, b in zip(y_at, x_at)], dtype=np.float32),
        "xp": xp, "yp": yp, "r_pow": np.array([lib.svo_powf(float(a), float(b)) for a, b in zip(xp, yp)], dtype=np.float32),
    }
    seeds = np.array([0, 1, 12345, 0xFFFFFFFF, 0x9E3779B9], dtype=np.uint32)
    seq = np.zeros((len(seeds), 16), dtype=np.uint32)
    uni = np.zeros((len(seeds), 16), dtype=np.float32)
    for i, sd in enumerate(seeds):
        st = (C.c_uint32 * 6)()
        lib.svo_xorwow_init(int(sd), st)
        for j in range(16):
            seq[i, j] = lib.svo_xorwow_next(st)
        lib.svo_xorwow_init(int(sd), st)
        for j in range(16):
            uni[i, j] = lib.svo_xorwow_uniform(st)
    kat.update({"rng_seeds": seeds, "rng_seq": seq, "rng_uniform": uni,
                "wang_in": np.arange(0, 64, dtype=np.uint32), "wang_out": np.array([lib.svo_wang_hash(i) for i in range(64)], dtype=np.uint32)})
    # texture fetches on the tiny_head scene
    sc = scenes.make_scene("tiny_head")
    o = binding.OracleScene(sc)
    uvw = rs.uniform(-0.05, 1.05, (128, 3)).astype(np.float32)
    kat["tex_uvw"] = uvw
    kat["tex3d"] = np.array([lib.svo_tex3d(o.ptr, float(a), float(b), float(c)) for a, b, c in uvw], dtype=np.float32)
    xs = rs.uniform(-0.1, 1.1, 64).astype(np.float32)
    t1 = np.zeros((64, 4), dtype=np.float32)
    for i, v in enumerate(xs):
        buf = (C.c_float * 4)()
        lib.svo_tex1d(o.ptr, float(v), buf)
        t1[i] = list(buf)
    kat["tex1d_x"], kat["tex1d"] = xs, t1
    uv = rs.uniform(-1.5, 2.5, (64, 2)).astype(np.float32)
    t2 = np.zeros((64, 4), dtype=np.float32)
    for i, (a, b) in enumerate(uv):
        buf = (C.c_float * 4)()
        lib.svo_tex2d(o.ptr, float(a), float(b), buf)
        t2[i] = list(buf)
    kat["tex2d_uv"], kat["tex2d"] = uv, t2
    np.savez_compressed(OUT / "kat.npz", **kat)

    # ---- rendered frames ----
    for name, depth, frames in RENDER_CASES:
        sc = scenes.make_scene(name, trace_depth=depth)
        hdr, img, cnt = oracle_frames(sc, frames)
        keys = sorted(cnt)
        np.savez_compressed(OUT / f"render_{name}_d{depth}_f{frames}.npz", hdr=hdr, img=img,
                            counter_names=np.array(keys), counters=np.array([cnt[k] for k in keys], dtype=np.uint64))
    sc = scenes.make_scene("tiny_head")
    rimg, rc = binding.OracleScene(sc).render_raycasting()
    np.savez_compressed(OUT / "raycast_tiny_head.npz", img=rimg, steps=np.uint64(rc["raycast_steps"]))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
