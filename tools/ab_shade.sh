# usage: tools/ab_shade.sh lib1.so lib2.so ...  (shade / extend kernel time per launch from the bench's own HIP events)
for l in "$@"; do
  echo "== $l"
  CRT_LIB_PATH=$PWD/computeraytracer_amd/lib/$l timeout -k 10 120 python bench.py --no-cpu-baseline --steps 6 --warmup 1 $AB_ARGS | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], 'Mrays/s  ms/step', d['ms_per_step'], 'ext_ms', r['launch_ms'], 'shade_ms', r['shade']['launch_ms'])" || exit 1
done
