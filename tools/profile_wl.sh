# usage (on the GPU box): bash tools/profile_wl.sh TAG WORKLOAD [bench args]  -> gpurun_out/TAG_wl_stats.md + bench line
set -e
tag=$1; wl=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_wprof -o vm --output-format csv -- python3 bench.py --no-cpu-baseline --workload $wl "$@" > gpurun_out/${tag}_wprof.log 2>&1
python3 tools/summarize_profile.py gpurun_out/${tag}_wprof/vm_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --workload $wl $*" > gpurun_out/${tag}_wl_stats.md
grep "^{" gpurun_out/${tag}_wprof.log | cut -c1-400
