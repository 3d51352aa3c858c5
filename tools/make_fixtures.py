#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the build container, where /root/reference exists).

  * monkey_smooth.mesh, cirno_low.mesh   the reference's OBJ assets baked with the build's own OBJ loader into the
                                         reference's MESH container (asset-baker conventions: de-indexed, uv.y = 1 - uy)
  * earthmap.tx                          assets/earthmap.png (2048x1024) box-filtered to 512x256, TEXI container
  * oracle_vectors.npz                   outputs of the CPU oracle on fixed inputs (regression pins; the reference itself
                                         has no tests or golden data -- "parity unpinned")
Nothing here copies reference source text; OBJ/PNG files are data and are stored re-encoded.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from computeraytracer_amd import abi  # noqa: E402
import oracle_lib as orc  # noqa: E402

REF = "/root/reference/assets"
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def bake_mesh(stem):
    p, n = C.c_void_p(), C.c_uint32()
    rc = abi.lib.crt_obj_to_vertices(os.path.join(REF, stem + ".obj").encode(), C.byref(p), C.byref(n))
    assert rc == 0, abi.lib.crt_host_last_error()
    verts = np.frombuffer((C.c_uint8 * (n.value * 44)).from_address(p.value), dtype=np.float32).copy()
    abi.lib.crt_free(p)
    idx = np.arange(n.value, dtype=np.uint32)
    rc = abi.lib.crt_mesh_file_write(os.path.join(OUT, stem + ".mesh").encode(), verts.ctypes.data, n.value, idx.ctypes.data,
                                     n.value, (stem + ".obj").encode())
    assert rc == 0
    print(stem, n.value // 3, "triangles ->", os.path.getsize(os.path.join(OUT, stem + ".mesh")), "bytes")


def bake_texture(src, dst, size):
    from PIL import Image
    im = Image.open(os.path.join(REF, src)).convert("RGBA").resize(size, Image.BOX)
    px = np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
    rc = abi.lib.crt_texture_file_write(os.path.join(OUT, dst).encode(), px.ctypes.data, size[0], size[1], src.encode())
    assert rc == 0
    print(dst, size, "->", os.path.getsize(os.path.join(OUT, dst)), "bytes")


def oracle_vectors():
    rng = np.random.default_rng(12345)
    out = {}
    # R1: hash outputs for (seed, x, y) triples
    seeds = np.concatenate([abi.frame_seeds(0, 16), np.float32([0.0, 1.5, 123.456, 2097151.9])])
    xs = rng.integers(0, 3840, 256).astype(np.uint32)
    ys = rng.integers(0, 2160, 256).astype(np.uint32)
    rows = []
    for i in range(256):
        s = C.c_float(orc.lib.orc_pixel_seed(float(seeds[i % len(seeds)]), int(xs[i]), int(ys[i])))
        o = (C.c_float * 3)()
        rec = [seeds[i % len(seeds)], xs[i], ys[i], s.value]
        for kind in (2, 2, 3, 1):
            orc.lib.orc_hash(C.byref(s), kind, o)
            rec += [o[0], o[1], o[2], s.value]
        rows.append(rec)
    out["hash_rows"] = np.array(rows, dtype=np.float64)
    # transcendental sequences
    th = np.linspace(0, 2 * np.pi, 1001).astype(np.float32)
    out["sincos_in"] = th
    out["sin_out"], out["cos_out"] = orc.math_fn(0, th)
    x01 = np.concatenate([np.float32([0, 2.0 ** -31, 1.0]), rng.random(1000).astype(np.float32)])
    out["cbrt_in"] = x01
    out["cbrt_out"] = orc.math_fn(1, x01)[0]
    xm = np.linspace(-1, 1, 1001).astype(np.float32)
    out["acos_in"] = xm
    out["acos_out"] = orc.math_fn(2, xm)[0]
    ay, ax = rng.standard_normal(1000).astype(np.float32), rng.standard_normal(1000).astype(np.float32)
    out["atan2_y"], out["atan2_x"] = ay, ax
    out["atan2_out"] = orc.math_fn(3, ay, ax)[0]
    # small renders: accumulation image + primary-hit ids
    for name, w, h, n in [("book1", 64, 36, 3), ("corne", 48, 48, 4), ("quads", 40, 40, 2), ("cirno", 64, 40, 2), ("monkey", 64, 36, 2)]:
        sc = abi.builtin_scene(name, w, h, OUT)
        seeds_n = abi.frame_seeds(0, n)
        img, ids, cnt = orc.render(sc, [sc.params_for(i + 1, seeds_n[i]) for i in range(n)], w, h, flags=orc.TRAVERSE_FAST)
        out[f"{name}_img"], out[f"{name}_ids"] = img, ids
        out[f"{name}_shape"] = np.array([w, h, n])
        out[f"{name}_rays"] = np.array([cnt.rays])
    np.savez_compressed(os.path.join(OUT, "oracle_vectors.npz"), **out)
    print("oracle_vectors.npz", os.path.getsize(os.path.join(OUT, "oracle_vectors.npz")), "bytes")


if __name__ == "__main__":
    bake_mesh("monkey_smooth")
    bake_mesh("cirno_low")
    bake_texture("earthmap.png", "earthmap.tx", (512, 256))
    oracle_vectors()
