# usage: tools/ab_g10.sh "ENV=.. ENV=.." ...   (config 5 frame time under different environment settings)
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py --workload grid10m_1080p --no-cpu-baseline --steps 8 --warmup 2 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$e:', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step build', d['config']['bvh_build_ms'])" || exit 1
done
