# usage: tools/ab_g10lib.sh lib1.so lib2.so ...  (config 5 frame time per library build; AB_ENV="K=V ..." optional)
for l in "$@"; do
  env CRT_LIB_PATH=$PWD/computeraytracer_amd/lib/$l $AB_ENV timeout -k 10 200 python bench.py --workload grid10m_1080p --no-cpu-baseline --steps 16 --warmup 14 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$l:', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step build', d['config']['bvh_build_ms'])" || exit 1
done
