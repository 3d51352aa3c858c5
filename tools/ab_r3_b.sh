# round 3: library variants, then while-while thresholds on the default build (headline workload)
bash tools/ab_lib.sh libcrt_w3.so libcrt.so libcrt_w7.so libcrt_w3.so libcrt.so libcrt_w7.so || exit 1
AB_ARGS="" bash tools/ab_env.sh "CRT_WF_LEAF_LANES=16" "CRT_WF_LEAF_LANES=20" "CRT_WF_LEAF_LANES=28" "CRT_WF_LEAF_LANES=32" "CRT_WF_REFILL=8" "CRT_WF_REFILL=12" "CRT_WF_REFILL=24" "CRT_WF_REFILL=32" ""
