"""Host side of the C ABI (no GPU): exported symbols, struct layouts, camera, scenes, asset formats."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from computeraytracer_amd import abi
import oracle_lib as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    text = open(os.path.join(ROOT, "include", "crt", "abi.h")).read()
    declared = set(re.findall(r"\b(crt_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 40
    lib = C.CDLL(abi.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert set(abi.EXPORTED) == declared
    assert b"gfx950" in abi.lib.crt_version()


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(abi.CrtError, match="no usable HIP device"):
        abi.Context(0)


def test_struct_sizes_match_reference_layouts():
    hdr = open(os.path.join(ROOT, "include", "crt", "types.h")).read()
    for name in ["crt_sphere", "crt_quad", "crt_tri", "crt_material", "crt_ref_bvh_node", "crt_camera_data", "crt_scene_params"]:
        assert f"}} {name};" in hdr
    assert (abi.sphere_dtype.itemsize, abi.quad_dtype.itemsize, abi.tri_dtype.itemsize) == (32, 96, 96)
    assert (abi.material_dtype.itemsize, abi.ref_node_dtype.itemsize, abi.params_dtype.itemsize) == (48, 64, 112)
    assert abi.params_dtype.fields["camera"][1] == 16 and abi.camera_dtype.fields["seed"][1] == 76


def test_camera_matches_oracle_restatement():
    for pos, at, fov, asp, ap, foc in [((10, 1.5, 2), (0, 0, -0.25), 30, 16 / 9, 1 / 45, 10), ((0, 0, 9), (0, 0, 0), 80, 1, 0, 10),
                                       ((1, 1, -2.878), (1, 1, 0), 40, 1, 0, 10)]:
        a = abi.camera(pos, at, fov, asp, ap, foc)
        b = np.zeros(1, abi.camera_dtype)
        p, t = np.float32(pos), np.float32(at)
        orc.lib.orc_camera(p.ctypes.data, t.ctypes.data, fov, asp, ap, foc, b.ctypes.data)
        assert a.tobytes() == b.tobytes()
        assert abs(np.dot(a["right"][0], a["up"][0])) < 1e-6 and a["iteration"][0] == 1.0


def test_builtin_scenes():
    s = abi.builtin_scene("book1", 400, 225)
    assert (len(s.spheres), len(s.quads), len(s.tris)) == (201, 0, 0) and len(s.world[2]) == 197 and s.world[2][-1] == 1
    assert s.spheres["radius"][-2] == np.float32(-0.9) and len(s.textures) == 1
    tex_mats = s.materials[s.materials["textureIndex"] != abi.BAD_INDEX]
    assert len(tex_mats) == 1 and tex_mats["type"][0] == 1
    s = abi.builtin_scene("corne", 64, 64)
    assert len(s.quads) == 18 and list(s.world[2]) == [0] * 6 + [1, 1] and (s.params["backgroundColor"] == 0).all()
    assert (s.materials["type"] == 8).sum() == 1 and len(s.materials) == 4   # de-duplicated by value (scene.cpp:9-29)
    s = abi.builtin_scene("voxel_city", 64, 36)
    assert len(s.tris) == 224998 and (s.params["backgroundColor"] == -1).all()
    uv = np.concatenate([s.tris["u"].ravel(), s.tris["v"].ravel()])
    assert uv.min() >= 0 and uv.max() <= 1
    s = abi.builtin_scene("grid:3", 64, 36)
    assert len(s.tris) == 9000 and s.world is None
    with pytest.raises(abi.CrtError, match="unknown scene"):
        abi.builtin_scene("nope", 8, 8)
    with pytest.raises(abi.CrtError):
        abi.builtin_scene("monkey", 8, 8, "/nonexistent")


def test_mesh_scenes_from_committed_fixtures(golden_dir):
    s = abi.builtin_scene("monkey", 64, 36, golden_dir)
    assert len(s.tris) == 968 and len(s.spheres) == 1
    s = abi.builtin_scene("cirno", 64, 36, golden_dir)
    assert len(s.tris) == 10000 and len(s.spheres) == 3 and s.textures[0].shape == (256, 512, 4)
    assert len(s.world[2]) == 10001


def test_quad_init_fields():
    q = abi.make_quad((1, 2, 3), (2, 0, 0), (0, 0, -4), 7)
    n = np.cross([2, 0, 0], [0, 0, -4]).astype(np.float32)
    assert np.allclose(q["normal"][0], n / np.linalg.norm(n)) and np.isclose(q["d"][0], np.dot(q["normal"][0], [1, 2, 3]))
    assert np.allclose(q["w"][0], n / np.dot(n, n)) and q["materialIndex"][0] == 7


def test_lz4_roundtrip_and_edge_cases():
    rng = np.random.default_rng(0)
    cases = [b"", b"a", b"abc" * 5, bytes(1000), bytes(rng.integers(0, 256, 5000, dtype=np.uint8)),
             (b"0123456789abcdef" * 300)[:4099], bytes(rng.integers(0, 4, 70000, dtype=np.uint8))]
    for data in cases:
        src = np.frombuffer(data, dtype=np.uint8).copy() if data else np.zeros(0, np.uint8)
        cap = abi.lib.crt_lz4_compress_bound(len(data))
        dst = np.zeros(cap, np.uint8)
        n = abi.lib.crt_lz4_compress(src.ctypes.data, len(data), dst.ctypes.data, cap)
        assert n > 0
        back = np.zeros(max(1, len(data)), np.uint8)
        m = abi.lib.crt_lz4_decompress(dst.ctypes.data, n, back.ctypes.data, len(data))
        assert m == len(data) and back[:m].tobytes() == data
    assert n < 56000   # 2-bit symbols: a greedy single-probe matcher still gains
    # a block written by the reference implementation's format: token 0x1f 'a' offset 1 len 15+4+... (hand-made RLE of 'a')
    blk = np.frombuffer(bytes([0x1F, ord("a"), 1, 0, 10, 0x50]) + b"bcdef", dtype=np.uint8).copy()
    out = np.zeros(64, np.uint8)
    m = abi.lib.crt_lz4_decompress(blk.ctypes.data, len(blk), out.ctypes.data, 64)
    assert m == 1 + 29 + 5 and out[:m].tobytes() == b"a" * 30 + b"bcdef"
    bad = np.frombuffer(bytes([0x0F, 9, 0]), dtype=np.uint8).copy()   # offset beyond output
    assert abi.lib.crt_lz4_decompress(bad.ctypes.data, 3, out.ctypes.data, 64) < 0


def test_obj_loader_and_mesh_container(tmp_path):
    obj = tmp_path / "t.obj"
    obj.write_text("# quad + tri\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 0.25\nvn 0 0 1\n"
                   "f 1/1/1 2/2/1 3/3/1 4/4/1\nf -4//1 -3//1 -2//1\nf 1 2 3\n")
    tris = abi.load_obj(str(obj), 5)
    assert len(tris) == 4 and (tris["materialIndex"] == 5).all()
    # fan triangulation of the quad: (1,2,3), (1,3,4); uv.y = 1 - uy (asset_main.cpp:48)
    assert list(tris["v2"][1]) == [0, 1, 0] and list(tris["u"][0]) == [0, 1, 1] and list(tris["v"][0]) == [1, 1, 0]
    assert tris["v"][1][2] == np.float32(0.75)
    assert list(tris["v1"][2]) == [1, 0, 0] and list(tris["v"][2]) == [1, 1, 1]    # no vt: uv = (0, 1 - 0)
    # bake to the reference's MESH container and read it back
    p, n = C.c_void_p(), C.c_uint32()
    assert abi.lib.crt_obj_to_vertices(str(obj).encode(), C.byref(p), C.byref(n)) == 0 and n.value == 12
    verts = np.frombuffer((C.c_uint8 * (n.value * 44)).from_address(p.value), dtype=np.float32).copy()
    abi.lib.crt_free(p)
    idx = np.arange(12, dtype=np.uint32)
    mesh = tmp_path / "t.mesh"
    assert abi.lib.crt_mesh_file_write(str(mesh).encode(), verts.ctypes.data, 12, idx.ctypes.data, 12, b"t.obj") == 0
    raw = mesh.read_bytes()
    assert raw[:4] == b"MESH" and int.from_bytes(raw[4:8], "little") == 1
    jl = int.from_bytes(raw[8:12], "little")
    import json
    meta = json.loads(raw[16:16 + jl])
    assert meta["vertex_format"] == "PNCV_F32" and meta["vertex_buffer_size"] == 12 * 44 and meta["index_buffer_size"] == 48
    assert meta["compression"] == "LZ4" and len(meta["bounds"]) == 7 and meta["index_size"] == 4
    pv, nv, pi, ni = C.c_void_p(), C.c_uint32(), C.c_void_p(), C.c_uint32()
    assert abi.lib.crt_mesh_file_read(str(mesh).encode(), C.byref(pv), C.byref(nv), C.byref(pi), C.byref(ni)) == 0
    back = np.frombuffer((C.c_uint8 * (nv.value * 44)).from_address(pv.value), dtype=np.float32)
    assert nv.value == 12 and ni.value == 12 and np.array_equal(back, verts)
    abi.lib.crt_free(pv); abi.lib.crt_free(pi)
    assert np.array_equal(abi.load_obj(str(mesh), 5), tris)
    with pytest.raises(abi.CrtError):
        abi.load_obj(str(tmp_path / "missing.obj"))


def test_texture_container(tmp_path):
    px = np.random.default_rng(1).integers(0, 256, (7, 5, 4), dtype=np.uint8)
    f = tmp_path / "t.tx"
    assert abi.lib.crt_texture_file_write(str(f).encode(), px.ctypes.data, 5, 7, b"t.png") == 0
    raw = f.read_bytes()
    assert raw[:4] == b"TEXI"
    p, w, h = C.c_void_p(), C.c_uint32(), C.c_uint32()
    assert abi.lib.crt_texture_file_read(str(f).encode(), C.byref(p), C.byref(w), C.byref(h)) == 0
    back = np.frombuffer((C.c_uint8 * (w.value * h.value * 4)).from_address(p.value), dtype=np.uint8).reshape(7, 5, 4)
    assert (w.value, h.value) == (5, 7) and np.array_equal(back, px)
    abi.lib.crt_free(p)
    assert abi.lib.crt_texture_file_read(str(tmp_path / "nope.tx").encode(), C.byref(p), C.byref(w), C.byref(h)) < 0


def test_texture_sampling_rule():
    # 2x2 texture, texel centres at +0.5, REPEAT: (0.25, 0.25) is the centre of texel (0,0); (0.5, 0.25) is halfway to (1,0)
    px = np.array([[[0, 0, 0, 255], [255, 0, 0, 255]], [[0, 255, 0, 255], [0, 0, 255, 255]]], dtype=np.uint8)
    out = (C.c_float * 3)()

    def s(u, v):
        orc.lib.orc_sample_texture(px.ctypes.data, 2, 2, u, v, out)
        return list(out)
    assert s(0.25, 0.25) == [0, 0, 0] and s(0.75, 0.25) == [1, 0, 0] and s(0.25, 0.75) == [0, 1, 0]
    assert s(0.5, 0.25) == [0.5, 0, 0]
    assert s(0.0, 0.25) == [0.5, 0, 0] and s(1.0, 0.25) == [0.5, 0, 0] and s(-0.75, 1.25) == [0, 0, 0]   # wrap


def test_band_rows_partition():
    for h in (1, 7, 8, 9, 64, 225, 2160):
        for world in (1, 2, 3, 8):
            rows = [abi.band_rows(h, r, world) for r in range(world)]
            allr = np.sort(np.concatenate(rows))
            assert np.array_equal(allr, np.arange(h))
            for r in range(world):
                assert all((y // 8) % world == r for y in rows[r])


def test_lost_empire_scene_uses_a_supplied_mesh(tmp_path):
    """Config 4 with the real mesh: absent here, so a tiny stand-in file under the same name checks the plumbing."""
    with pytest.raises(abi.CrtError):
        abi.builtin_scene("lost_empire", 64, 36, str(tmp_path))          # no mesh: an error, not a silent substitute
    (tmp_path / "lost_empire.obj").write_text("v 0 0 0\nv 4 0 0\nv 4 2 0\nv 0 2 8\nvt 0 0\nvt 1 0\nvt 1 1\nf 1/1 2/2 3/3\nf 1/1 3/3 4/2\n")
    s = abi.builtin_scene("lost_empire", 64, 36, str(tmp_path))
    assert len(s.tris) == 2 and len(s.spheres) == 0 and len(s.materials) == 1
    assert int(s.materials["type"][0]) == 1 and int(s.materials["textureIndex"][0]) == 0 and len(s.textures) == 1
    cam = s.params["camera"][0]
    assert np.isfinite(cam["position"]).all() and cam["position"][1] > 1.0       # above the box, looking at its centre
    offs, prims, is_list = s.world
    assert list(offs) == [0, 1, 2] and not is_list.any()


def test_asset_readers_reject_lying_headers_and_long_obj_lines(tmp_path):
    """sizes read from a file must not drive allocations or casts unchecked; OBJ face lines may be long."""
    import struct
    # a header that claims 4 GiB of JSON + blob in a 16-byte file
    bad = tmp_path / "huge.mesh"
    bad.write_bytes(b"MESH" + struct.pack("<III", 1, 0xFFFFFFF0, 0xFFFFFFF0))
    p, n, q, m = C.c_void_p(), C.c_uint32(), C.c_void_p(), C.c_uint32()
    assert abi.lib.crt_mesh_file_read(os.fsencode(str(bad)), C.byref(p), C.byref(n), C.byref(q), C.byref(m)) == -4
    assert b"claims more bytes" in abi.lib.crt_host_last_error()
    # negative / fractional / non-finite buffer sizes in the JSON
    for vbs in ("-44", "1e300", "43", "nan"):
        js = ('{"compression":"None","index_buffer_size":0,"index_size":4,"original_file":"x","vertex_buffer_size":%s,"vertex_format":"PNCV_F32"}' % vbs).encode()
        f = tmp_path / "bad.mesh"
        f.write_bytes(b"MESH" + struct.pack("<III", 1, len(js), 0) + js)
        assert abi.lib.crt_mesh_file_read(os.fsencode(str(f)), C.byref(p), C.byref(n), C.byref(q), C.byref(m)) == -4, vbs
    js = b'{"buffer_size":-4,"compression":"None","format":"RGBA8","height":1,"original_file":"x","width":-1}'
    t = tmp_path / "bad.tx"
    t.write_bytes(b"TEXI" + struct.pack("<III", 1, len(js), 0) + js)
    w, h = C.c_uint32(), C.c_uint32()
    assert abi.lib.crt_texture_file_read(os.fsencode(str(t)), C.byref(p), C.byref(w), C.byref(h)) == -4
    # an OBJ polygon whose 'f' line is far longer than any fixed line buffer: 2000 corners -> 1998 fan triangles, none dropped
    k = 2000
    ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
    obj = tmp_path / "fan.obj"
    obj.write_text("".join(f"v {np.cos(a):.6f} {np.sin(a):.6f} 0\n" for a in ang) + "f " + " ".join(str(i + 1) for i in range(k)) + "\n")
    assert os.path.getsize(obj) > 8000
    tris = abi.load_obj(str(obj))
    assert len(tris) == k - 2
    assert np.allclose(tris["v0"], [1, 0, 0]) and np.allclose(tris["v2"][-1], [np.cos(ang[-1]), np.sin(ang[-1]), 0], atol=1e-6)


def test_missing_rccl_is_an_error_code_not_a_crash():
    """load_rccl called dlerror() twice (the second call returns NULL) and built a std::string from it, so a box
    without librccl segfaulted instead of reporting.  CRT_RCCL_SONAME points the loader at a library that does not exist;
    a child process, because the loader runs once per process."""
    import subprocess
    import sys
    code = (
        "import os, sys\n"
        "sys.path.insert(0, %r)\n"
        "os.environ['CRT_RCCL_SONAME'] = 'libcrt_no_such_rccl.so.9'\n"
        "from computeraytracer_amd import abi\n"
        "import ctypes as C\n"
        "buf = (C.c_uint8 * abi.COMM_ID_BYTES)()\n"
        "rc = abi.lib.crt_comm_unique_id(buf)\n"
        "msg = abi.lib.crt_host_last_error().decode()\n"
        "print(rc, '|', msg)\n"
        "rc2 = abi.lib.crt_comm_unique_id(buf)\n"
        "print(rc2)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    first, second = r.stdout.strip().splitlines()
    rc, msg = first.split(" | ", 1)
    assert int(rc) == -2 and int(second) == -2
    assert "RCCL is not loadable" in msg and "libcrt_no_such_rccl.so.9" in msg


def test_lz4_decoder_against_real_liblz4_blocks():
    """(f)2 pinned outside the repo: tests/golden/lz4_vectors.npz holds LZ4_compress_default outputs of the real liblz4
    (tools/make_fixtures.py, which also asserts that LZ4_decompress_safe restores every crt_lz4_compress block).  The
    in-tree decoder must restore every plaintext, at exact capacity, and refuse a capacity one byte short."""
    v = np.load(os.path.join(ROOT, "tests", "golden", "lz4_vectors.npz"), allow_pickle=False)
    plain, po, block, bo = v["plain"], v["plain_off"], v["block"], v["block_off"]
    assert len(po) == len(bo) and len(po) - 1 >= 20
    assert bytes(v["liblz4_version"]).decode().startswith("1.")
    for i in range(len(po) - 1):
        want = plain[po[i]:po[i + 1]]
        blk = np.ascontiguousarray(block[bo[i]:bo[i + 1]])
        out = np.full(len(want) + 8, 0xAB, np.uint8)
        n = abi.lib.crt_lz4_decompress(blk.ctypes.data, len(blk), out.ctypes.data, len(want))
        assert n == len(want), (i, n, len(want))
        assert np.array_equal(out[:n], want) and (out[len(want):] == 0xAB).all(), i
        if len(want) > 0:
            assert abi.lib.crt_lz4_decompress(blk.ctypes.data, len(blk), out.ctypes.data, len(want) - 1) < 0, i
        # the in-tree compressor's block for the same plaintext decodes to it as well (the real decoder's verdict on
        # these blocks was taken when the vectors were made)
        src = np.ascontiguousarray(want) if len(want) else np.zeros(1, np.uint8)
        cap = abi.lib.crt_lz4_compress_bound(len(want))
        mine = np.zeros(max(cap, 16), np.uint8)
        k = abi.lib.crt_lz4_compress(src.ctypes.data, len(want), mine.ctypes.data, cap)
        assert k > 0
        back = np.zeros(max(len(want), 1), np.uint8)
        assert abi.lib.crt_lz4_decompress(mine.ctypes.data, k, back.ctypes.data, len(want)) == len(want)
        assert np.array_equal(back[:len(want)], want)
