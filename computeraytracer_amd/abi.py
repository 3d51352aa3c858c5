"""ctypes binding of the C ABI in include/crt/abi.h (libcrt.so).

This is plumbing for the Python harnesses (tests, bench.py, the multi-GPU
driver); the product is the shared library.  There is no CPU fallback: if
libcrt.so is missing, import fails; if no GPU is usable, `Context()` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRT_LIB_PATH") or os.path.join(_HERE, "lib", "libcrt.so")   # override: A/B builds of the same ABI

BAD_INDEX = 0xFFFFFFFF
ID_MISS = 0xFFFFFFFF
KIND_SPHERE, KIND_QUAD, KIND_TRI = 0, 1, 2
RNG_REF, RNG_PCG = 0, 1
TRACER_MEGAKERNEL, TRACER_WAVEFRONT = 0, 1
FLAG_FULL_IMAGE = 1
FLAG_DEBUG_SMALL_STACK = 2
FLAG_PER_BOUNCE_KERNELS = 4
FLAG_PATH_KERNEL = 8
FLAG_PATH_KERNEL_PERSISTENT = 16
BVH_LBVH, BVH_SAH_REFINE, BVH_LEAF_COLLAPSE, BVH_WIDE, BVH_WIDE_DIRECT = 0, 1, 2, 4, 8
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT = 1, 2, 4, 8     # crt_material.type (rt_material.h:9-14)
EXT_TEXTURE_INDEX, EXT_ALPHA_TEST = 1, 2     # crt_extensions_set (build extensions, off by default)

# numpy views of the POD layouts in include/crt/types.h
sphere_dtype = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("pad0", "<f4", 3), ("materialIndex", "<u4")])
quad_dtype = np.dtype([("corner", "<f4", 3), ("d", "<f4"), ("u", "<f4", 3), ("pad0", "<f4"), ("v", "<f4", 3), ("pad1", "<f4"),
                       ("normal", "<f4", 3), ("pad2", "<f4"), ("w", "<f4", 3), ("pad3", "<f4"), ("pad4", "<f4", 3),
                       ("materialIndex", "<u4")])
tri_dtype = np.dtype([("v0", "<f4", 3), ("pad0", "<f4"), ("v1", "<f4", 3), ("pad1", "<f4"), ("v2", "<f4", 3), ("pad2", "<f4"),
                      ("u", "<f4", 3), ("pad3", "<f4"), ("v", "<f4", 3), ("pad4", "<f4"), ("pad5", "<f4", 3),
                      ("materialIndex", "<u4")])
material_dtype = np.dtype([("albedo", "<f4", 3), ("fuzziness", "<f4"), ("pad0", "<f4", 3), ("type", "<u4"),
                           ("pad1", "<f4", 3), ("textureIndex", "<u4")])
ray_dtype = np.dtype([("origin", "<f4", 3), ("t_max", "<f4"), ("direction", "<f4", 3), ("pad", "<u4")])
ray_hit_dtype = np.dtype([("t", "<f4"), ("id", "<u4")])
QUERY_CLOSEST, QUERY_ANY = 0, 1
camera_dtype = np.dtype([("position", "<f4", 3), ("shouldRenderAABB", "<u4"), ("backward", "<f4", 3), ("lensRadius", "<f4"),
                         ("right", "<f4", 3), ("focusDistance", "<f4"), ("up", "<f4", 3), ("iteration", "<f4"),
                         ("horizontal", "<f4", 3), ("seed", "<f4"), ("vertical", "<f4", 3), ("pad", "<f4")])
params_dtype = np.dtype([("backgroundColor", "<f4", 3), ("pad", "<f4"), ("camera", camera_dtype)])
bvh_node_dtype = np.dtype([("lo", "<f4", 3), ("link", "<u4"), ("hi", "<f4", 3), ("count", "<u4")])
ref_node_dtype = np.dtype([("min", "<f4", 3), ("pad1", "<f4"), ("max", "<f4", 3), ("pad2", "<f4"), ("objectIndex", "<u4"),
                           ("hitIndex", "<u4"), ("missIndex", "<u4"), ("pad3", "<f4"), ("type", "<u4"),
                           ("numChildren", "<u4"), ("pad4", "<f4", 2)])
assert sphere_dtype.itemsize == 32 and quad_dtype.itemsize == 96 and tri_dtype.itemsize == 96
assert material_dtype.itemsize == 48 and params_dtype.itemsize == 112 and bvh_node_dtype.itemsize == 32
assert ref_node_dtype.itemsize == 64


class Texture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("spheres", C.c_void_p), ("n_spheres", C.c_uint32),
                ("quads", C.c_void_p), ("n_quads", C.c_uint32),
                ("tris", C.c_void_p), ("n_tris", C.c_uint32),
                ("materials", C.c_void_p), ("n_materials", C.c_uint32),
                ("textures", C.POINTER(Texture)), ("n_textures", C.c_uint32)]


class Config(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rng_mode", C.c_uint32), ("tracer", C.c_uint32),
                ("flags", C.c_uint32), ("max_bounces", C.c_uint32), ("band_rank", C.c_uint32), ("band_world", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("node_visits", C.c_uint64), ("prim_tests", C.c_uint64),
                ("last_render_ms", C.c_double), ("last_build_ms", C.c_double), ("last_extend_ms", C.c_double),
                ("last_extend_launches", C.c_uint64), ("last_shade_ms", C.c_double), ("last_shade_launches", C.c_uint64),
                ("last_primary_ms", C.c_double), ("last_primary_launches", C.c_uint64)]


class CrtError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C computeraytracer_amd/csrc` "
                          "(or __graft_entry__.build()); there is no fallback path")
    lib = C.CDLL(LIB_PATH)
    vp, u32, i32, f32 = C.c_void_p, C.c_uint32, C.c_int, C.c_float
    P = C.POINTER
    sig = {
        "crt_version": (C.c_char_p, []),
        "crt_create": (i32, [i32, P(vp)]),
        "crt_destroy": (None, [vp]),
        "crt_last_error": (C.c_char_p, [vp]),
        "crt_configure": (i32, [vp, P(Config)]),
        "crt_scene_upload": (i32, [vp, P(SceneDesc)]),
        "crt_extensions_set": (i32, [vp, u32]),
        "crt_bvh_build": (i32, [vp, u32, P(f32)]),
        "crt_bvh_refit": (i32, [vp, P(f32)]),
        "crt_poll": (i32, [vp, P(i32)]),
        "crt_retune": (i32, [vp]),
        "crt_debug_sort_u64": (i32, [vp, vp, u32, u32, u32]),
        "crt_debug_sort_pairs_u64": (i32, [vp, vp, vp, u32, u32, u32]),
        "crt_debug_math": (i32, [vp, i32, u32, vp, vp, vp, vp]),
        "crt_tris_update": (i32, [vp, vp, u32, u32]),
        "crt_tris_animate": (i32, [vp, f32, f32]),
        "crt_tris_read": (i32, [vp, vp, u32, u32]),
        "crt_dispatch": (i32, [vp, vp]),
        "crt_render": (i32, [vp, vp, u32, u32, vp]),
        "crt_rays_query": (i32, [vp, vp, u32, u32, vp]),
        "crt_rays_query_noalloc": (i32, [vp, vp, u32, u32, vp]),
        "crt_sync": (i32, [vp]),
        "crt_image_read": (i32, [vp, vp, C.c_size_t]),
        "crt_hit_ids_read": (i32, [vp, vp, C.c_size_t]),
        "crt_image_device_ptr": (i32, [vp, P(vp), P(C.c_size_t)]),
        "crt_image_copy_to_device": (i32, [vp, vp, C.c_size_t]),
        "crt_image_export_fd": (i32, [vp, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
        "crt_band_rows": (u32, [u32, u32, u32, vp]),
        "crt_comm_unique_id": (i32, [vp]),
        "crt_comm_init": (i32, [vp, vp, u32, u32]),
        "crt_comm_destroy": (i32, [vp]),
        "crt_image_gather": (i32, [vp, vp]),
        "crt_image_gathered_read": (i32, [vp, vp, C.c_size_t]),
        "crt_bands_assemble": (i32, [vp, vp, u32, u32, u32, vp]),
        "crt_stats_read": (i32, [vp, P(Stats)]),
        "crt_stats_reset": (i32, [vp]),
        "crt_set_instrumented": (i32, [vp, i32]),
        "crt_set_kernel_timing": (i32, [vp, i32]),
        "crt_loop_counters_read": (i32, [vp, vp, u32]),
        "crt_bvh_read": (i32, [vp, vp, P(u32), vp, vp, vp]),
        "crt_bvh_wide_read": (i32, [vp, vp, P(u32), P(u32)]),
        "crt_debug_boxes_upload": (i32, [vp, vp, u32]),
        "crt_camera_init": (None, [vp, P(f32), P(f32), f32, f32, f32, f32]),
        "crt_frame_seeds": (None, [u32, u32, vp]),
        "crt_quad_init": (None, [vp, P(f32), P(f32), P(f32), u32]),
        "crt_scene_builtin": (i32, [C.c_char_p, C.c_char_p, u32, u32, P(vp)]),
        "crt_host_scene_desc": (P(SceneDesc), [vp]),
        "crt_host_scene_params": (vp, [vp]),
        "crt_host_scene_world": (i32, [vp, P(u32), P(vp), P(vp), P(vp)]),
        "crt_scene_free": (None, [vp]),
        "crt_host_last_error": (C.c_char_p, []),
        "crt_load_obj": (i32, [C.c_char_p, u32, P(vp), P(u32)]),
        "crt_free": (None, [vp]),
        "crt_mesh_file_read": (i32, [C.c_char_p, P(vp), P(u32), P(vp), P(u32)]),
        "crt_mesh_file_write": (i32, [C.c_char_p, vp, u32, vp, u32, C.c_char_p]),
        "crt_texture_file_read": (i32, [C.c_char_p, P(vp), P(u32), P(u32)]),
        "crt_texture_file_write": (i32, [C.c_char_p, vp, u32, u32, C.c_char_p]),
        "crt_obj_to_vertices": (i32, [C.c_char_p, P(vp), P(u32)]),
        "crt_lz4_compress_bound": (i32, [i32]),
        "crt_lz4_compress": (i32, [vp, i32, vp, i32]),
        "crt_lz4_decompress": (i32, [vp, i32, vp, i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sig)


lib, EXPORTED = _load()


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def frame_seeds(first, count):
    out = np.zeros(count, dtype=np.float32)
    lib.crt_frame_seeds(first, count, _ptr(out))
    return out


def camera(position, at, fov=70.0, aspect=1.0, aperture=1.0 / 45.0, focus=10.0):
    out = np.zeros(1, dtype=camera_dtype)
    lib.crt_camera_init(_ptr(out), _f3(position), _f3(at), fov, aspect, aperture, focus)
    return out


def make_quad(corner, u, v, material):
    out = np.zeros(1, dtype=quad_dtype)
    lib.crt_quad_init(_ptr(out), _f3(corner), _f3(u), _f3(v), material)
    return out


COMM_ID_BYTES = 128


def code_rev():
    """A short hash of the tracer's kernel sources (trace.hip and the headers it includes).  The committed counter files
    under profiles/ carry the value they were collected with; bench.py marks their figures stale when it differs."""
    import hashlib
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "hip")
    h = hashlib.sha256()
    for f in ("trace.hip", "intersect.h", "intersect_wide.h", "shade.h", "device_math.h", "build_internal.h"):
        try:
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
        except OSError:
            return None
    return h.hexdigest()[:16]


def comm_unique_id():
    """crt_comm_unique_id: the opaque 128-byte id rank 0 distributes to the other ranks."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    rc = lib.crt_comm_unique_id(buf)
    if rc != 0:
        raise CrtError(f"crt_comm_unique_id failed ({rc}): {lib.crt_host_last_error().decode()}")
    return bytes(buf)


def band_rows(height, rank, world):
    n = lib.crt_band_rows(height, rank, world, None)
    rows = np.zeros(n, dtype=np.uint32)
    lib.crt_band_rows(height, rank, world, _ptr(rows))
    return rows


class HostScene:
    """Arrays of one scene in the reference's layouts (numpy, owned by Python)."""

    def __init__(self, spheres=None, quads=None, tris=None, materials=None, textures=(), params=None, world=None, name=""):
        self.name = name
        self.spheres = np.zeros(0, sphere_dtype) if spheres is None else np.ascontiguousarray(spheres, dtype=sphere_dtype)
        self.quads = np.zeros(0, quad_dtype) if quads is None else np.ascontiguousarray(quads, dtype=quad_dtype)
        self.tris = np.zeros(0, tri_dtype) if tris is None else np.ascontiguousarray(tris, dtype=tri_dtype)
        self.materials = np.zeros(0, material_dtype) if materials is None else np.ascontiguousarray(materials, dtype=material_dtype)
        self.textures = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]   # each (h, w, 4)
        self.params = np.zeros(1, params_dtype) if params is None else params
        self.world = world   # (obj_offsets u32[n+1], obj_prims u32[], obj_is_list u8[n]) or None

    @property
    def n_prims(self):
        return len(self.spheres) + len(self.quads) + len(self.tris)

    def desc(self):
        """ctypes crt_scene_desc pointing into the numpy arrays (keep `self` alive while it is used)."""
        self._tex_structs = (Texture * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            self._tex_structs[i].rgba8 = t.ctypes.data
            self._tex_structs[i].width = t.shape[1]
            self._tex_structs[i].height = t.shape[0]
        d = SceneDesc()
        d.spheres, d.n_spheres = _ptr(self.spheres), len(self.spheres)
        d.quads, d.n_quads = _ptr(self.quads), len(self.quads)
        d.tris, d.n_tris = _ptr(self.tris), len(self.tris)
        d.materials, d.n_materials = _ptr(self.materials), len(self.materials)
        d.textures = C.cast(self._tex_structs, C.POINTER(Texture)) if self.textures else None
        d.n_textures = len(self.textures)
        return d

    def params_for(self, iteration, seed):
        p = self.params.copy()
        p["camera"]["iteration"] = np.float32(iteration)
        p["camera"]["seed"] = np.float32(seed)
        return p


def _copy_array(ptr, n, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype)
    buf = (C.c_uint8 * (n * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def builtin_scene(name, width, height, asset_dir=""):
    """One of the built-in scenes (crt_scene_builtin), copied into numpy arrays."""
    h = C.c_void_p()
    rc = lib.crt_scene_builtin(name.encode(), (asset_dir or "").encode(), width, height, C.byref(h))
    if rc != 0:
        raise CrtError(f"crt_scene_builtin({name!r}) failed ({rc}): {lib.crt_host_last_error().decode()}")
    try:
        d = lib.crt_host_scene_desc(h).contents
        textures = []
        for i in range(d.n_textures):
            t = d.textures[i]
            px = _copy_array(t.rgba8, t.width * t.height * 4, np.dtype(np.uint8))
            textures.append(px.reshape(t.height, t.width, 4))
        params = _copy_array(lib.crt_host_scene_params(h), 1, params_dtype)
        n_obj = C.c_uint32()
        po, pp, pl = C.c_void_p(), C.c_void_p(), C.c_void_p()
        lib.crt_host_scene_world(h, C.byref(n_obj), C.byref(po), C.byref(pp), C.byref(pl))
        world = None
        if n_obj.value:
            offs = _copy_array(po.value, n_obj.value + 1, np.dtype("<u4"))
            world = (offs, _copy_array(pp.value, int(offs[-1]), np.dtype("<u4")), _copy_array(pl.value, n_obj.value, np.dtype(np.uint8)))
        return HostScene(_copy_array(d.spheres, d.n_spheres, sphere_dtype), _copy_array(d.quads, d.n_quads, quad_dtype),
                         _copy_array(d.tris, d.n_tris, tri_dtype), _copy_array(d.materials, d.n_materials, material_dtype),
                         textures, params, world, name)
    finally:
        lib.crt_scene_free(h)


def load_obj(path, material_index=0):
    p, n = C.c_void_p(), C.c_uint32()
    rc = lib.crt_load_obj(os.fsencode(path), material_index, C.byref(p), C.byref(n))
    if rc != 0:
        raise CrtError(f"crt_load_obj failed ({rc}): {lib.crt_host_last_error().decode()}")
    try:
        return _copy_array(p.value, n.value, tri_dtype)
    finally:
        lib.crt_free(p)


class Context:
    """One GPU context (crt_create .. crt_destroy)."""

    def __init__(self, device=0):
        h = C.c_void_p()
        rc = lib.crt_create(device, C.byref(h))
        if rc != 0:
            raise CrtError(f"crt_create(device={device}) failed ({rc}): no usable HIP device; there is no CPU fallback")
        self._h = h
        self.cfg = None
        self._scene = None

    def close(self):
        if self._h:
            lib.crt_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise CrtError(f"{what} failed ({rc}): {lib.crt_last_error(self._h).decode()}")

    def configure(self, width, height, rng_mode=RNG_REF, tracer=TRACER_MEGAKERNEL, flags=0, max_bounces=0, band_rank=0, band_world=1):
        self.cfg = Config(width, height, rng_mode, tracer, flags, max_bounces, band_rank, band_world)
        self._check(lib.crt_configure(self._h, C.byref(self.cfg)), "crt_configure")

    def upload(self, scene):
        self._scene = scene
        d = scene.desc()
        self._check(lib.crt_scene_upload(self._h, C.byref(d)), "crt_scene_upload")

    def extensions_set(self, ext_flags):
        self._check(lib.crt_extensions_set(self._h, ext_flags), "crt_extensions_set")

    def build(self, flags=BVH_LBVH, wait=True):
        """crt_bvh_build.  wait=False passes build_ms = NULL: a CRT_BVH_WIDE_DIRECT rebuild then only enqueues (abi.h) and
        the build's time is in stats().last_build_ms after the next sync; returns None."""
        if not wait:
            self._check(lib.crt_bvh_build(self._h, flags, None), "crt_bvh_build")
            return None
        ms = C.c_float()
        self._check(lib.crt_bvh_build(self._h, flags, C.byref(ms)), "crt_bvh_build")
        return ms.value

    def refit(self, want_cost=False):
        """crt_bvh_refit: boxes of the existing 8-wide tree from the primitives as they are now; want_cost waits and returns
        sum(area(node)) / area(root)."""
        if not want_cost:
            self._check(lib.crt_bvh_refit(self._h, None), "crt_bvh_refit")
            return None
        c = C.c_float()
        self._check(lib.crt_bvh_refit(self._h, C.byref(c)), "crt_bvh_refit")
        return c.value

    def debug_sort_u64(self, keys, bit_lo, bit_hi):
        k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
        self._check(lib.crt_debug_sort_u64(self._h, _ptr(k), len(k), bit_lo, bit_hi), "crt_debug_sort_u64")
        return k

    def debug_sort_pairs_u64(self, keys, vals, bit_lo, bit_hi):
        k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
        v = np.ascontiguousarray(vals, dtype=np.uint32).copy()
        self._check(lib.crt_debug_sort_pairs_u64(self._h, _ptr(k), _ptr(v), len(k), bit_lo, bit_hi), "crt_debug_sort_pairs_u64")
        return k, v

    def debug_math(self, fn, a, b=None):
        """crt_debug_math: (out0, out1) of function `fn` on the device for float32 arrays a (and b)."""
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), dtype=np.float32)
        o0, o1 = np.zeros_like(a), np.zeros_like(a)
        self._check(lib.crt_debug_math(self._h, fn, a.size, _ptr(a), _ptr(b), _ptr(o0), _ptr(o1)), "crt_debug_math")
        return o0, o1

    def retune(self):
        self._check(lib.crt_retune(self._h), "crt_retune")

    def poll(self):
        """crt_poll: True when everything enqueued so far has finished."""
        d = C.c_int()
        self._check(lib.crt_poll(self._h, C.byref(d)), "crt_poll")
        return bool(d.value)

    def animate(self, frame, amplitude):
        self._check(lib.crt_tris_animate(self._h, frame, amplitude), "crt_tris_animate")

    def tris_update(self, tris, first=0):
        t = np.ascontiguousarray(tris, dtype=tri_dtype)
        self._check(lib.crt_tris_update(self._h, _ptr(t), first, len(t)), "crt_tris_update")

    def tris_read(self, first=0, count=None):
        n = len(self._scene.tris) - first if count is None else count
        out = np.zeros(n, dtype=tri_dtype)
        self._check(lib.crt_tris_read(self._h, _ptr(out), first, n), "crt_tris_read")
        return out

    def dispatch(self, params):
        p = np.ascontiguousarray(params, dtype=params_dtype)
        self._check(lib.crt_dispatch(self._h, _ptr(p)), "crt_dispatch")

    def render(self, params, first_iteration, seeds):
        p = np.ascontiguousarray(params, dtype=params_dtype)
        s = np.ascontiguousarray(seeds, dtype=np.float32)
        self._check(lib.crt_render(self._h, _ptr(p), first_iteration, len(s), _ptr(s)), "crt_render")

    def rays_query(self, rays, mode=QUERY_CLOSEST):
        """crt_rays_query on a numpy array of ray_dtype; returns an array of ray_hit_dtype."""
        r = np.ascontiguousarray(rays, dtype=ray_dtype)
        out = np.zeros(len(r), dtype=ray_hit_dtype)
        self._check(lib.crt_rays_query(self._h, _ptr(r) if len(r) else None, len(r), mode, _ptr(out) if len(r) else None), "crt_rays_query")
        return out

    def rays_query_noalloc(self, rays_dev_ptr, n_rays, mode, hits_dev_ptr):
        """crt_rays_query_noalloc: raw device pointers (ints) in, nothing returned; enqueues only."""
        self._check(lib.crt_rays_query_noalloc(self._h, C.c_void_p(rays_dev_ptr), n_rays, mode, C.c_void_p(hits_dev_ptr)), "crt_rays_query_noalloc")

    def sync(self):
        self._check(lib.crt_sync(self._h), "crt_sync")

    def image(self):
        out = np.zeros((self.cfg.height, self.cfg.width, 4), dtype=np.float32)
        self._check(lib.crt_image_read(self._h, _ptr(out), out.size), "crt_image_read")
        return out

    def hit_ids(self):
        out = np.zeros((self.cfg.height, self.cfg.width), dtype=np.uint32)
        self._check(lib.crt_hit_ids_read(self._h, _ptr(out), out.size), "crt_hit_ids_read")
        return out

    def image_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(lib.crt_image_device_ptr(self._h, C.byref(p), C.byref(n)), "crt_image_device_ptr")
        return p.value, n.value

    def image_copy_to_device(self, device_ptr, n_bytes):
        self._check(lib.crt_image_copy_to_device(self._h, C.c_void_p(device_ptr), n_bytes), "crt_image_copy_to_device")

    def image_export_fd(self, want_fd=True):
        """crt_image_export_fd: enqueues the copy of the band buffer into the exportable allocation; returns (fd or None,
        image_bytes, alloc_bytes).  The caller owns the descriptor (os.close it, or hand it to an importer that takes it over)."""
        fd, nb, ab = C.c_int(-1), C.c_size_t(), C.c_size_t()
        self._check(lib.crt_image_export_fd(self._h, C.byref(fd) if want_fd else None, C.byref(nb), C.byref(ab)), "crt_image_export_fd")
        return (fd.value if want_fd else None), nb.value, ab.value

    def comm_init(self, comm_id, rank, world):
        """crt_comm_init: comm_id = the 128 bytes rank 0 got from comm_unique_id()."""
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(comm_id))
        self._check(lib.crt_comm_init(self._h, buf, rank, world), "crt_comm_init")

    def comm_destroy(self):
        self._check(lib.crt_comm_destroy(self._h), "crt_comm_destroy")

    def image_gather(self, full_dev_ptr=None):
        self._check(lib.crt_image_gather(self._h, C.c_void_p(full_dev_ptr) if full_dev_ptr else None), "crt_image_gather")

    def image_gathered(self):
        out = np.zeros((self.cfg.height, self.cfg.width, 4), dtype=np.float32)
        self._check(lib.crt_image_gathered_read(self._h, _ptr(out), out.size), "crt_image_gathered_read")
        return out

    def bands_assemble(self, gathered_dev_ptr, width, height, world, full_dev_ptr):
        self._check(lib.crt_bands_assemble(self._h, C.c_void_p(gathered_dev_ptr), width, height, world, C.c_void_p(full_dev_ptr)), "crt_bands_assemble")

    def stats(self):
        s = Stats()
        self._check(lib.crt_stats_read(self._h, C.byref(s)), "crt_stats_read")
        return s

    def stats_reset(self):
        self._check(lib.crt_stats_reset(self._h), "crt_stats_reset")

    def set_instrumented(self, on):
        self._check(lib.crt_set_instrumented(self._h, int(bool(on))), "crt_set_instrumented")

    def loop_counters(self):
        out = np.zeros((16, 6), dtype=np.uint64)
        self._check(lib.crt_loop_counters_read(self._h, _ptr(out), 96), "crt_loop_counters_read")
        return out

    def set_kernel_timing(self, on):
        self._check(lib.crt_set_kernel_timing(self._h, int(bool(on))), "crt_set_kernel_timing")

    def bvh(self):
        """(nodes, sphere_ids, quad_ids, tri_ids): ids are full primitive IDs (kind << 30 | index)."""
        n = C.c_uint32()
        self._check(lib.crt_bvh_read(self._h, None, C.byref(n), None, None, None), "crt_bvh_read")
        nodes = np.zeros(n.value, dtype=bvh_node_dtype)
        sc = self._scene
        s = np.zeros(len(sc.spheres), dtype=np.uint32)
        q = np.zeros(len(sc.quads), dtype=np.uint32)
        t = np.zeros(len(sc.tris), dtype=np.uint32)
        self._check(lib.crt_bvh_read(self._h, _ptr(nodes), C.byref(n), _ptr(s), _ptr(q), _ptr(t)), "crt_bvh_read")
        return nodes, s | np.uint32(KIND_SPHERE << 30), q | np.uint32(KIND_QUAD << 30), t | np.uint32(KIND_TRI << 30)

    def bvh_wide(self):
        """(nodes as uint32[n, 32], levels): the 8-wide quantised nodes of a CRT_BVH_WIDE build (bvh_wide.hip)."""
        n, lv = C.c_uint32(), C.c_uint32()
        self._check(lib.crt_bvh_wide_read(self._h, None, C.byref(n), C.byref(lv)), "crt_bvh_wide_read")
        nodes = np.zeros((n.value, 32), dtype=np.uint32)
        self._check(lib.crt_bvh_wide_read(self._h, _ptr(nodes), C.byref(n), C.byref(lv)), "crt_bvh_wide_read")
        return nodes, lv.value

    def debug_boxes_upload(self, ref_nodes):
        """ref_nodes: numpy array of ref_node_dtype (the reference's BVHNode array) for the AABB-outline debug view."""
        ref_nodes = np.ascontiguousarray(ref_nodes, dtype=ref_node_dtype)
        self._keep_boxes = ref_nodes
        self._check(lib.crt_debug_boxes_upload(self._h, _ptr(ref_nodes), len(ref_nodes)), "crt_debug_boxes_upload")
