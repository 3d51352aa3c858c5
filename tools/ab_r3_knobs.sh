# usage: tools/ab_r3_knobs.sh   (round 3: zero-code experiments on the headline workload; whole-step time, no per-kernel events)
run() {  # $1 = env string, rest = bench args
  e="$1"; shift
  echo "== env: $e args: $*"
  env CRT_BENCH_NO_KERNEL_TIMING=1 $e timeout -k 10 150 python bench.py --no-cpu-baseline "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], 'Mrays/s  ms/step', d['ms_per_step'], 'spp', d['config']['spp_per_step'])" || exit 1
}
run "" --steps 8 --warmup 2
run "CRT_WF_TAIL_PATHS=1e9 CRT_WF_TAIL_BOUNCE=0 CRT_WF_PATH_PERSIST=1" --steps 4 --warmup 1
run "CRT_WF_TAIL_PATHS=1e9 CRT_WF_TAIL_BOUNCE=1 CRT_WF_PATH_PERSIST=1" --steps 4 --warmup 1
run "CRT_WF_TAIL_PATHS=1e9 CRT_WF_TAIL_BOUNCE=4 CRT_WF_PATH_PERSIST=1" --steps 8 --warmup 2
run "CRT_WF_TAIL_PATHS=1e9 CRT_WF_TAIL_BOUNCE=6 CRT_WF_PATH_PERSIST=1" --steps 8 --warmup 2
run "CRT_WF_TAIL_PATHS=1e9 CRT_WF_TAIL_BOUNCE=4" --steps 8 --warmup 2
run "CRT_WF_TAIL_PATHS=1e9 CRT_WF_TAIL_BOUNCE=6" --steps 8 --warmup 2
run "CRT_WF_TAIL_PATHS=1e9 CRT_WF_TAIL_BOUNCE=8" --steps 8 --warmup 2
run "CRT_WF_BATCH_PATHS=270e6" --steps 4 --warmup 1 --spp-per-step 32
run "CRT_WF_BATCH_PATHS=540e6" --steps 2 --warmup 1 --spp-per-step 64
