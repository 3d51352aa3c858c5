# usage (on the GPU box): bash tools/profile_build.sh TAG [workload] [bvh]  -> gpurun_out/TAG_build_stats.md
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_bprof -o vm --output-format csv -- python3 tools/build_only.py ${2:-lost_empire_4k} ${3:-sah-wide} 5 > gpurun_out/${tag}_bprof.log 2>&1
python3 tools/summarize_profile.py gpurun_out/${tag}_bprof/vm_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 tools/build_only.py ${2:-lost_empire_4k} ${3:-sah-wide} 5" > gpurun_out/${tag}_build_stats.md
grep "build ms" gpurun_out/${tag}_bprof.log
