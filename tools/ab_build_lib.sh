# usage (GPU box): bash tools/ab_build_lib.sh lib1.so lib2.so ...  -- static build time and per-kernel stats of library variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for l in "$@"; do
  echo "== $l"
  CRT_LIB_PATH=$GRAFT_REPO_ROOT/computeraytracer_amd/lib/$l timeout -k 10 120 python3 tools/build_only.py lost_empire_4k sah-wide 9 || exit 1
  CRT_LIB_PATH=$GRAFT_REPO_ROOT/computeraytracer_amd/lib/$l timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/abb_$l -o vm --output-format csv -- python3 tools/build_only.py lost_empire_4k sah-wide 5 > gpurun_out/abb_$l.log 2>&1 || exit 1
  python3 tools/summarize_profile.py gpurun_out/abb_$l/vm_kernel_stats.csv "$l" | grep "k_sah_group\|k_dp_level"
done
