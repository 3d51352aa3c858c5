# usage (GPU box): bash tools/prof_lib.sh TAG lib.so [bench args]  -> gpurun_out/TAG_kernel_stats.md (rocprofv3 --kernel-trace --stats of bench.py with that build of libcrt)
set -e
tag=$1; lib=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CRT_LIB_PATH=$PWD/computeraytracer_amd/lib/$lib
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof -o vm --output-format csv -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" > gpurun_out/${tag}_prof.log 2>&1
python3 tools/summarize_profile.py gpurun_out/${tag}_prof/vm_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 $* (libcrt = $lib)" > gpurun_out/${tag}_kernel_stats.md
head -14 gpurun_out/${tag}_kernel_stats.md
