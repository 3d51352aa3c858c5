run() { timeout -k 10 120 python bench.py --no-cpu-baseline --steps 16 --warmup 2 "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; print(d['value'], 'Mrays/s  V', c['nodes_per_ray'], 'T', c['prim_tests_per_ray'], 'ext_ms', d['roofline']['launch_ms'], 'build', c['bvh_build_ms'])"; }
for l in 8 16 24 32 40 48; do echo "== leaf_lanes=$l"; CRT_WF_LEAF_LANES=$l run || exit 1; done
for r in 8 24 32; do echo "== refill=$r leaf=32"; CRT_WF_LEAF_LANES=32 CRT_WF_REFILL=$r run || exit 1; done
