import os, sys, time
sys.path.insert(0, os.getcwd())
from computeraytracer_amd import abi
W, H = 3840, 2160
scene = abi.builtin_scene("voxel_city", W, H)
for world in (1, 2, 4, 8):
    ctx = abi.Context(0)
    ctx.configure(W, H, tracer=abi.TRACER_WAVEFRONT, band_rank=0, band_world=world)
    ctx.upload(scene); ctx.build(abi.BVH_SAH_REFINE | abi.BVH_WIDE)
    seeds = abi.frame_seeds(0, 200)
    ctx.render(scene.params, 1, seeds[:64]); ctx.sync(); ctx.stats_reset()
    t0 = time.perf_counter(); ctx.render(scene.params, 65, seeds[64:192]); ctx.sync(); dt = time.perf_counter() - t0
    st = ctx.stats()
    print(f"world {world}: rank0 {st.rays / dt / 1e6:8.1f} Mrays/s  x{world} = {world * st.rays / dt / 1e6:8.1f}  ms/step {dt / 128 * 1e3:.3f}", flush=True)
    ctx.close()
