run() { timeout -k 10 120 python bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; print(d['value'], 'Mrays/s  ms/step', d['ms_per_step'], 'ext_ms', d['roofline']['launch_ms'])"; }
echo "== base"; run
for g in 2048 4096 8192; do echo "== s8 shade grid=$g"; CRT_LIB_PATH=$PWD/computeraytracer_amd/lib/libcrt_s8.so CRT_WF_SHADE_GRID=$g run || exit 1; done
