# usage (on the GPU box): bash tools/collect_profile.sh TAG   -> gpurun_out/TAG_{bench.json,kernel_stats.md,pmc.json}
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py --steps 8 --warmup 1 > gpurun_out/${tag}_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof -o vm --output-format csv -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 1 > gpurun_out/${tag}_prof.log 2>&1
python3 tools/summarize_profile.py gpurun_out/${tag}_prof/vm_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 1 (wavefront tracer, 8-wide BVH, voxel_city 4K; a step = 16 iterations = one batch, so every k_wf_extend<false,...> launch has the size of the timed ones; the <true,...> launches are the one instrumented step)" > gpurun_out/${tag}_kernel_stats.md
bash tools/pmc_passes.sh gpurun_out/${tag}_pmc > gpurun_out/${tag}_pmc.log 2>&1 || { tail -5 gpurun_out/${tag}_pmc.log; exit 1; }
python3 tools/pmc_to_json.py gpurun_out/${tag}_pmc > gpurun_out/${tag}_pmc.json
