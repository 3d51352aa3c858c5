# usage: tools/ab_lib.sh lib1.so lib2.so ...   (A/B builds of the same ABI on the headline workload, SAH tree; kernel times from the bench line)
for l in "$@"; do
  echo "== $l"
  CRT_LIB_PATH=$PWD/computeraytracer_amd/lib/$l timeout -k 10 120 python bench.py --no-cpu-baseline --steps 4 --warmup 1 $AB_ARGS | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; r=d['roofline']; print(d['value'], 'Mrays/s  ms/step', d['ms_per_step'], 'V', c['nodes_per_ray'], 'T', c['prim_tests_per_ray'], 'ext_ms', r['launch_ms'], 'prim_ms', r.get('primary',{}).get('launch_ms'), 'shade_ms', r.get('shade',{}).get('launch_ms'))" || exit 1
done
