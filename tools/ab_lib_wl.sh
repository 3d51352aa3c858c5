# usage: tools/ab_lib_wl.sh WORKLOAD lib1.so lib2.so ...   (A/B builds of the same ABI on one bench workload)
wl=$1; shift
for l in "$@"; do
  echo "== $wl $l"
  CRT_LIB_PATH=$PWD/computeraytracer_amd/lib/$l timeout -k 10 120 python bench.py --no-cpu-baseline --workload $wl --steps 8 --warmup 1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; print(d['value'], 'Mrays/s  build_ms', c['bvh_build_ms'], 'V', c['nodes_per_ray'], 'T', c['prim_tests_per_ray'], 'ms/step', d['ms_per_step'])" || exit 1
done
