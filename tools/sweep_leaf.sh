for ll in 8 12 16 20 32 40; do echo "leaf_lanes=$ll"; CRT_WF_LEAF_LANES=$ll bash tools/ab_lib.sh libcrt.so | tail -1; done
