#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the build container, where /root/reference exists).

  * monkey_smooth.mesh, cirno_low.mesh   the reference's OBJ assets baked with the build's own OBJ loader into the
                                         reference's MESH container (asset-baker conventions: de-indexed, uv.y = 1 - uy)
  * earthmap.tx                          assets/earthmap.png (2048x1024) box-filtered to 512x256, TEXI container
  * oracle_vectors.npz                   outputs of the CPU oracle on fixed inputs (regression pins; the reference itself
                                         has no tests or golden data -- "parity unpinned")
  * lz4_vectors.npz                      real LZ4 blocks: LZ4_compress_default outputs of the system liblz4 (1.9.3 in the build
                                         container, /opt/conda/lib/liblz4.so.1) for ~20 buffers, with their plaintexts.  The
                                         reference calls LZ4_compress_default / LZ4_decompress_safe (asset-lib/src/mesh_asset.cpp:51,106,
                                         texture_asset.cpp:35,67); the in-tree codec has to decode what the real one writes, and the
                                         real one has to decode what the in-tree one writes (asserted here, at generation time: the
                                         library itself never travels to the GPU box, the vectors do)
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


LZ4_SO = "/opt/conda/lib/liblz4.so.1"


def lz4_vectors():
    lz = C.CDLL(LZ4_SO)
    lz.LZ4_versionString.restype = C.c_char_p
    lz.LZ4_compressBound.argtypes = [C.c_int]
    lz.LZ4_compress_default.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lz.LZ4_decompress_safe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    rng = np.random.default_rng(4242)
    mesh = open(os.path.join(OUT, "monkey_smooth.mesh"), "rb").read()
    hdr = np.frombuffer(mesh[:16], dtype=np.uint32)
    blob = np.frombuffer(mesh[16 + int(hdr[2]):16 + int(hdr[2]) + int(hdr[3])], dtype=np.uint8)
    # the monkey's vertex blob, decoded by the in-tree reader and cut to 40 KiB of PNCV floats
    p, n, q, m = C.c_void_p(), C.c_uint32(), C.c_void_p(), C.c_uint32()
    assert abi.lib.crt_mesh_file_read(os.path.join(OUT, "monkey_smooth.mesh").encode(), C.byref(p), C.byref(n), C.byref(q), C.byref(m)) == 0
    verts = bytes((C.c_uint8 * (n.value * 44)).from_address(p.value))
    abi.lib.crt_free(p); abi.lib.crt_free(q)
    # ... and the real library agrees with the in-tree reader about that blob
    back = np.zeros(n.value * 44 + m.value * 4, np.uint8)
    got = lz.LZ4_decompress_safe(blob.ctypes.data, back.ctypes.data, len(blob), len(back))
    assert got == len(back) and bytes(back[:n.value * 44]) == verts, "liblz4 and the in-tree reader disagree about monkey_smooth.mesh"
    words = (b"the quick brown fox jumps over the lazy dog " * 40)
    bufs = [
        b"", b"\x00", b"a", b"ab" * 6, b"abcdefghijkl", b"abcdefghijklm", bytes(13), bytes(64), bytes(65536),
        bytes(range(256)) * 8, words, words[:257], b"x" * 4096 + b"y",
        rng.integers(0, 256, 1000, dtype=np.uint8).tobytes(), rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(),
        rng.integers(0, 4, 5000, dtype=np.uint8).tobytes(), rng.integers(0, 2, 70001, dtype=np.uint8).tobytes(),
        np.arange(5000, dtype=np.uint32).tobytes(), verts[:40960], verts[40960:40960 + 12345],
        np.tile(rng.integers(0, 256, 300, dtype=np.uint8), 300).tobytes(),          # matches at distance 300, long runs of them
        (bytes(70000) + b"tail"),                                                  # a match longer than 65535
    ]
    plain, comp, p_off, c_off = [], [], [0], [0]
    for b in bufs:
        src = np.frombuffer(b, dtype=np.uint8).copy() if b else np.zeros(1, np.uint8)
        cap = lz.LZ4_compressBound(len(b))
        dst = np.zeros(max(cap, 16), np.uint8)
        k = lz.LZ4_compress_default(src.ctypes.data, dst.ctypes.data, len(b), len(dst))
        assert k > 0, (len(b), k)
        # the in-tree compressor's output must be a block the real decoder accepts and restores
        cap2 = abi.lib.crt_lz4_compress_bound(len(b))
        mine = np.zeros(max(cap2, 16), np.uint8)
        k2 = abi.lib.crt_lz4_compress(src.ctypes.data, len(b), mine.ctypes.data, cap2)
        assert k2 > 0, (len(b), k2)
        back = np.zeros(max(len(b), 1), np.uint8)
        got = lz.LZ4_decompress_safe(mine.ctypes.data, back.ctypes.data, k2, len(b))
        assert got == len(b) and bytes(back[:len(b)]) == b, f"liblz4 rejects crt_lz4_compress output for a {len(b)}-byte buffer"
        plain.append(np.frombuffer(b, dtype=np.uint8)); comp.append(dst[:k].copy())
        p_off.append(p_off[-1] + len(b)); c_off.append(c_off[-1] + k)
    np.savez_compressed(os.path.join(OUT, "lz4_vectors.npz"),
                        plain=np.concatenate(plain), plain_off=np.array(p_off, np.int64),
                        block=np.concatenate(comp), block_off=np.array(c_off, np.int64),
                        liblz4_version=np.frombuffer(lz.LZ4_versionString(), dtype=np.uint8))
    print("lz4_vectors.npz", len(bufs), "buffers, liblz4", lz.LZ4_versionString().decode(), os.path.getsize(os.path.join(OUT, "lz4_vectors.npz")), "bytes;",
          "liblz4 decodes every crt_lz4_compress block")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "lz4":      # only the LZ4 vectors (the other fixtures are byte-stable and already committed)
        lz4_vectors()
        sys.exit(0)
    bake_mesh("monkey_smooth")
    bake_mesh("cirno_low")
    bake_texture("earthmap.png", "earthmap.tx", (512, 256))
    oracle_vectors()
    lz4_vectors()
