# usage: tools/ab_tail.sh  (one-sample dispatches at several frame sizes, tail kernel off / from bounce 0, 1, 2)
for wl in "cirno_1080p 1" "monkey_1080p 1" "lost_empire_4k 1" "cirno_1080p 2" "cirno_1080p 4"; do
  set -- $wl
  for tb in -1 0 1 2; do
    CRT_WF_TAIL_PATHS=1e9 CRT_WF_TAIL_BOUNCE=$tb timeout -k 10 200 python bench.py --workload $1 --spp-per-step $2 --no-cpu-baseline --steps 12 --warmup 3 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1 spp $2 tail $tb:', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step')" || exit 1
  done
done
