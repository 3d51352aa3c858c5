# usage (on the GPU box): bash tools/kernel_times.sh TAG -- <bench.py arguments>   -> per-kernel average times of that bench run
tag=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_kt -o vm --output-format csv -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/${tag}_kt.log 2>&1 || { tail -5 gpurun_out/${tag}_kt.log; exit 1; }
python3 tools/summarize_profile.py gpurun_out/${tag}_kt/vm_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline $*" > gpurun_out/${tag}_kernel_stats.md
head -16 gpurun_out/${tag}_kernel_stats.md
