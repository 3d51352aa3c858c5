set -e
for v in "sah 4" "lbvh-wide 4" "sah-wide 4"; do
  set -- $v
  echo "== bvh=$1 max_leaf=$2"
  CRT_BVH_MAX_LEAF=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 4 --warmup 1 --bvh $1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; print(d['value'], 'Mrays/s  build_ms', c['bvh_build_ms'], 'V', c['nodes_per_ray'], 'T', c['prim_tests_per_ray'], 'ext_ms', d['roofline']['launch_ms'])"
done
