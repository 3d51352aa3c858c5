# usage (GPU box): bash tools/prof_dispatch.sh [workload]  -> kernel timeline of the last unbatched dispatch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/dprof -o vm --output-format csv -- python3 tools/dispatch_time.py ${1:-cirno_1080p} 8 > gpurun_out/dprof.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/dprof/vm_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last=rows[-8:]
t0=int(last[0]['Start_Timestamp'])
for r in last:
    print(r['Kernel_Name'][:48].ljust(48), round((int(r['Start_Timestamp'])-t0)/1e3,1), round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1))
PY
