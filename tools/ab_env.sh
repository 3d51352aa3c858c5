# usage: tools/ab_env.sh "VAR=val ..." "VAR=val ..." ...   (A/B of environment knobs on the headline workload; "" = defaults)
for e in "$@"; do
  echo "== env: $e"
  env $e timeout -k 10 120 python bench.py --no-cpu-baseline --steps 4 --warmup 1 $AB_ARGS | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; print(d['value'], 'Mrays/s  ms/step', d['ms_per_step'], 'build_ms', c['bvh_build_ms'], 'V', c['nodes_per_ray'], 'T', c['prim_tests_per_ray'], 'ext_ms', d['roofline']['launch_ms'])" || exit 1
done
