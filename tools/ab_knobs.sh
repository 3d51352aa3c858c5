run() { timeout -k 10 120 python bench.py --no-cpu-baseline --steps 32 --warmup 16 "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; print(d['value'], 'Mrays/s  ms/step', d['ms_per_step'], 'ext_ms', d['roofline']['launch_ms'])"; }
for s in 32 64; do echo "== shards=$s"; CRT_WF_SHARDS=$s run || exit 1; done
for t in 67e6; do echo "== batch paths=$t"; CRT_WF_BATCH_PATHS=$t run || exit 1; done
for g in 3072 6144; do echo "== shade grid=$g"; CRT_WF_SHADE_GRID=$g run || exit 1; done
