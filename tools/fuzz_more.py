#!/usr/bin/env python3
"""More seeds of tests/test_gpu_fuzz.py than the suite runs: python tools/fuzz_more.py FIRST COUNT [scenes|configs|queries|ext|soups|souprender|soupasync|textures|scales|updates|poison|rays|contexts] (on a GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from computeraytracer_amd import abi
import test_gpu_fuzz as t

first, count = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "scenes"
fn = {"configs": t.test_random_configurations_bit_exact, "queries": t.test_random_ray_queries_from_random_origins, "ext": t.test_random_scenes_with_build_extensions, "soups": t.test_random_soups_of_random_size_on_every_tree, "souprender": t.test_random_soups_rendered_animated_and_refitted, "soupasync": t.test_random_soups_rebuilt_without_waiting, "textures": t.test_random_texture_sizes, "scales": t.test_random_scenes_at_random_scales_and_offsets, "updates": t.test_random_partial_triangle_updates, "poison": t.test_random_scenes_with_nan_and_infinity_in_the_primitives, "rays": t.test_random_degenerate_rays, "contexts": (lambda _ctx, seed: t.test_two_contexts_interleaved_on_one_gpu(seed)), "scenes": t.test_random_scenes_bit_exact_on_every_tree_and_tracer}[mode]
ctx = abi.Context(0)
bad = 0
for seed in range(first, first + count):
    if (seed - first) % (20 if mode.startswith("soup") else 500) == 0:
        print("at seed", seed, "failures so far", bad, flush=True)
    try:
        fn(ctx, seed)
    except AssertionError as e:
        bad += 1
        import traceback
        print("seed", seed, "FAILED", str(e)[:200], traceback.format_exc().splitlines()[-3].strip(), flush=True)
print("seeds", first, "..", first + count - 1, "failures", bad, flush=True)
ctx.close()
sys.exit(1 if bad else 0)
