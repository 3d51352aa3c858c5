"""SAH cost of the device's binary BVH next to a CPU full-sweep top-down SAH build of the same triangles (the classic
quality yardstick): how much tree quality the LBVH + refinement pipeline leaves on the table.

  python tools/sah_gap.py device [workload]     (GPU box: builds sah, reads the tree back, prints its cost)
  python tools/sah_gap.py sweep  [workload]     (CPU only, minutes: full-sweep SAH, leaves <= 4, same cost constants)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from computeraytracer_amd import abi  # noqa: E402
import bench  # noqa: E402

C_T, C_I, MAX_LEAF = 1.0, 1.0, 4


def half_area(lo, hi):
    d = np.maximum(hi - lo, 0.0)
    return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]


def device_cost(wl):
    scene_name, W, H, _ = bench.WORKLOADS[wl]
    scene = abi.builtin_scene(scene_name, W, H, os.path.join(ROOT, "tests", "golden"))
    ctx = abi.Context(0)
    ctx.configure(64, 64)
    ctx.upload(scene)
    for name, flags in (("lbvh", abi.BVH_LBVH), ("sah", abi.BVH_SAH_REFINE)):
        ctx.build(flags)
        nodes = ctx.bvh()[0]
        lo, hi = nodes["lo"].astype(np.float64), nodes["hi"].astype(np.float64)
        a = half_area(lo, hi)
        root = half_area(np.minimum(lo[0], lo[1]), np.maximum(hi[0], hi[1]))
        # only what is reachable from pair 0 (a rebuilt tree uses fewer pairs than the array holds)
        reach = np.zeros(len(nodes), bool)
        front = np.array([0], np.int64)
        while len(front):
            kids = np.concatenate([2 * front, 2 * front + 1])
            reach[kids] = True
            front = nodes["link"][kids[nodes["count"][kids] == 0]].astype(np.int64)
        inner = (nodes["count"] == 0) & reach
        a = np.where(reach, a, 0.0)
        n = (nodes["count"] & 0x0FFFFFFF).astype(np.float64)
        cost = C_T + (C_T * a[inner].sum() + C_I * (a[~inner] * n[~inner]).sum()) / root      # + the root's own visit
        leaves = (~inner) & (n > 0)
        print(f"{name}: SAH cost {cost:.3f}  inner {inner.sum()}  leaves {leaves.sum()}  prims/leaf {n[leaves].mean():.2f}", flush=True)
    ctx.close()


def sweep_cost(wl):
    scene_name, W, H, _ = bench.WORKLOADS[wl]
    scene = abi.builtin_scene(scene_name, W, H, os.path.join(ROOT, "tests", "golden"))
    assert len(scene.spheres) == 0 and len(scene.quads) == 0, "triangles only"
    t = scene.tris
    v = np.stack([t["v0"], t["v1"], t["v2"]], 1).astype(np.float64)
    plo, phi = v.min(1), v.max(1)
    cen = 0.5 * (plo + phi)
    n_all = len(plo)
    root = half_area(plo.min(0), phi.max(0))
    total, n_inner, n_leaf, leaf_prims = 0.0, 0, 0, 0
    stack = [np.arange(n_all)]
    t0 = last = time.time()
    done = 0
    while stack:
        idx = stack.pop()
        n = len(idx)
        lo, hi = plo[idx], phi[idx]
        area = half_area(lo.min(0), hi.max(0))
        best, best_axis, best_k, best_order = np.inf, -1, -1, None
        if n > 1:
            for ax in range(3):
                order = np.argsort(cen[idx, ax], kind="stable")
                l, h = lo[order], hi[order]
                la = half_area(np.minimum.accumulate(l, 0), np.maximum.accumulate(h, 0))[:-1]
                ra = half_area(np.minimum.accumulate(l[::-1], 0), np.maximum.accumulate(h[::-1], 0))[::-1][1:]
                k = np.arange(1, n)
                c = la * k + ra * (n - k)
                j = int(np.argmin(c))
                if c[j] < best:
                    best, best_axis, best_k, best_order = c[j], ax, j + 1, order
        split_cost = C_T * area + C_I * best
        if n <= MAX_LEAF and (n == 1 or C_I * area * n <= split_cost):
            total += C_I * area * n
            n_leaf += 1
            leaf_prims += n
            done += n
        else:
            total += C_T * area
            n_inner += 1
            s = idx[best_order]
            stack.append(s[:best_k])
            stack.append(s[best_k:])
        if time.time() - last > 30:
            last = time.time()
            print(f"  {done}/{n_all} primitives in leaves, {time.time() - t0:.0f} s", flush=True)
    # the device tree's inner-node sum does not include the root box itself (its two children are nodes 0 and 1); add it there instead
    print(f"sweep: SAH cost {total / root:.3f}  inner {n_inner}  leaves {n_leaf}  prims/leaf {leaf_prims / n_leaf:.2f}", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1]
    wl = sys.argv[2] if len(sys.argv) > 2 else "lost_empire_4k"
    (device_cost if mode == "device" else sweep_cost)(wl)
