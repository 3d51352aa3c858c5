"""Random scenes for tests/test_gpu_fuzz.py (GPU against the oracle) and tests/test_oracle_bvh.py (the oracle's tree against
its brute force): every primitive kind, every material kind, a texture, lens blur, either background mode."""
import numpy as np

from computeraytracer_amd import abi


def random_scene(seed, w, h):
    rs = np.random.RandomState(1000 + seed)
    n_mat = 7
    mats = np.zeros(n_mat, abi.material_dtype)
    mats["albedo"] = rs.uniform(0.05, 0.95, (n_mat, 3)).astype(np.float32)
    mats["type"] = rs.choice([abi.MAT_LAMBERTIAN, abi.MAT_METAL, abi.MAT_DIELECTRIC, abi.MAT_DIFFUSE_LIGHT], n_mat, p=[0.45, 0.25, 0.15, 0.15])
    mats["type"][0] = abi.MAT_LAMBERTIAN
    mats["fuzziness"] = np.where(mats["type"] == abi.MAT_DIELECTRIC, rs.uniform(1.1, 2.4, n_mat), rs.uniform(0.0, 0.6, n_mat)).astype(np.float32)
    mats["albedo"][mats["type"] == abi.MAT_DIFFUSE_LIGHT] *= np.float32(8.0)
    mats["textureIndex"] = 0xFFFFFFFF
    mats["textureIndex"][0] = 0                                   # one textured Lambertian
    tex = rs.randint(0, 256, (8, 16, 4)).astype(np.uint8)
    n_s, n_q, n_t = [int(rs.randint(0, m)) for m in (30, 12, 400)]
    if seed % 3 == 0:
        n_t += 300                                                # more than one SAH group now and then
    spheres = np.zeros(n_s, abi.sphere_dtype)
    spheres["center"] = rs.uniform(-4, 4, (n_s, 3)).astype(np.float32) * np.float32([1, 0.4, 1])
    spheres["radius"] = (rs.uniform(0.1, 0.9, n_s) * rs.choice([1.0, 1.0, 1.0, -1.0], n_s)).astype(np.float32)      # negative: the normal flips (vk_engine.cpp:853)
    spheres["materialIndex"] = rs.randint(0, n_mat, n_s)
    quads = np.zeros(n_q, abi.quad_dtype)
    for i in range(n_q):
        quads[i] = abi.make_quad(rs.uniform(-4, 4, 3), rs.uniform(-1.5, 1.5, 3), rs.uniform(-1.5, 1.5, 3), int(rs.randint(0, n_mat)))[0]
    tris = np.zeros(n_t, abi.tri_dtype)
    c = rs.uniform(-4, 4, (n_t, 3)).astype(np.float32) * np.float32([1, 0.5, 1])
    e = rs.uniform(-0.7, 0.7, (2, n_t, 3)).astype(np.float32)
    tris["v0"], tris["v1"], tris["v2"] = c, c + e[0], c + e[1]
    tris["u"] = rs.uniform(-1, 2, (n_t, 3)).astype(np.float32); tris["v"] = rs.uniform(-1, 2, (n_t, 3)).astype(np.float32)
    tris["materialIndex"] = rs.randint(0, n_mat, n_t)
    if n_t > 20:
        tris["v2"][:5] = tris["v1"][:5]                           # a few degenerate ones
    params = np.zeros(1, abi.params_dtype)
    params["backgroundColor"] = (-1.0, -1.0, -1.0) if seed % 2 == 0 else tuple(rs.uniform(0, 0.4, 3))      # sky gradient / flat colour
    pos = rs.uniform(-6, 6, 3) * np.float32([1, 0.3, 1]) + np.float32([0, 1.0, 0])
    if seed == 5 and n_s:
        pos = spheres["center"][0] + np.float32([0.01, 0.02, 0.0])                                        # the camera inside a sphere
    params["camera"] = abi.camera(tuple(pos), tuple(rs.uniform(-1, 1, 3)), float(rs.uniform(30, 80)), w / h, float(rs.choice([0.0, 1.0 / 45.0, 0.2])), float(rs.uniform(3, 9)))
    return abi.HostScene(spheres=spheres, quads=quads, tris=tris, materials=mats, textures=[tex], params=params, name=f"fuzz-{seed}")


def random_soup(seed):
    """A triangle soup of random size (a few hundred to ~80 k triangles) in random clusters of random spread, some spheres and
    quads among them: tree SHAPES nobody chose (SAH tiers of one, two and three levels, lopsided groups, deep collapses)."""
    rs = np.random.RandomState(50000 + seed)
    n_t = int(np.exp(rs.uniform(np.log(200), np.log(80000))))
    n_c = int(rs.randint(1, 40))
    centres = rs.uniform(-20, 20, (n_c, 3)).astype(np.float32) * np.float32([1, 0.3, 1])
    spread = np.exp(rs.uniform(np.log(0.05), np.log(6.0), n_c)).astype(np.float32)
    size = np.exp(rs.uniform(np.log(0.01), np.log(1.5), n_c)).astype(np.float32)
    k = rs.randint(0, n_c, n_t)
    c = (centres[k] + rs.standard_normal((n_t, 3)).astype(np.float32) * spread[k, None]).astype(np.float32)
    e = (rs.uniform(-1, 1, (2, n_t, 3)).astype(np.float32) * size[k, None]).astype(np.float32)
    tris = np.zeros(n_t, abi.tri_dtype)
    tris["v0"], tris["v1"], tris["v2"] = c, c + e[0], c + e[1]
    if seed % 4 == 0:                                              # many copies of the same triangle: equal Morton codes, equal boxes
        tris[: n_t // 3] = tris[0]
    tris["materialIndex"] = 0
    base = random_scene(seed, 64, 64)
    n_s, n_q = (int(rs.randint(0, 40)), int(rs.randint(0, 20))) if seed % 2 else (0, 0)
    spheres = np.zeros(n_s, abi.sphere_dtype)
    spheres["center"] = rs.uniform(-20, 20, (n_s, 3)).astype(np.float32) * np.float32([1, 0.3, 1])
    spheres["radius"] = rs.uniform(0.1, 3.0, n_s).astype(np.float32)
    quads = np.zeros(n_q, abi.quad_dtype)
    for i in range(n_q):
        quads[i] = abi.make_quad(rs.uniform(-20, 20, 3) * [1, 0.3, 1], rs.uniform(-4, 4, 3), rs.uniform(-4, 4, 3), 0)[0]
    return abi.HostScene(spheres=spheres, quads=quads, tris=tris, materials=base.materials, textures=base.textures, params=base.params, name=f"soup-{seed}")


def soup_rays(scene, seed, n):
    rs = np.random.RandomState(seed)
    rays = np.zeros(n, abi.ray_dtype)
    # Every ray is aimed at (or near) the centre of a random triangle from an origin 10 to 300 triangle sizes away.  Origins
    # thousands of sizes away are avoided on purpose: the shader's determinant test has an ABSOLUTE threshold, so a tiny
    # triangle seen from far away at a shallow angle has barycentrics that are rounding noise, "hits" can lie outside the
    # triangle's box and then depend on the tree (the first version of this test met three such rays in
    # 2 x 10^7, each among exact duplicates, where the tie-break makes them visible).
    pick = rs.randint(0, len(scene.tris), n)
    tv = scene.tris[pick]
    centre = ((tv["v0"] + tv["v1"]) + tv["v2"]) / np.float32(3.0)
    size = np.maximum(np.linalg.norm(tv["v1"] - tv["v0"], axis=1), np.linalg.norm(tv["v2"] - tv["v0"], axis=1))[:, None].astype(np.float32)
    target = centre + rs.standard_normal((n, 3)).astype(np.float32) * size * np.where(rs.randint(0, 2, (n, 1)) == 0, np.float32(2.0), np.float32(0.0))
    dirn = rs.standard_normal((n, 3)).astype(np.float32)
    dirn /= np.linalg.norm(dirn, axis=1, keepdims=True).astype(np.float32)
    rays["origin"] = (target - dirn * size * np.exp(rs.uniform(np.log(10.0), np.log(300.0), (n, 1))).astype(np.float32)).astype(np.float32)
    d = (target - rays["origin"]).astype(np.float32)
    rays["direction"] = d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-20).astype(np.float32)
    rays["t_max"] = np.inf
    return rays
