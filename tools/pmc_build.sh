# usage (GPU box): bash tools/pmc_build.sh outdir [workload] [bvh] -- SQ/TA/TCC counters per kernel for BVH builds only
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $out/p$i -o vm -- python3 tools/build_only.py ${2:-grid10m_1080p} ${3:-lbvh-wide} 2 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
  echo "pass $i ok"
done
python3 - $out <<'PY'
import csv, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for p in (1, 2):
    for r in csv.DictReader(open(f"{out}/p{p}/vm_counter_collection.csv")):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        agg[k]["_n%d" % p] += 0
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    g = c.get("GRBM_GUI_ACTIVE", 0)
    if g < 1e4: continue
    print(f"{k:42s} gui {g/1e3:9.0f}k  valu_busy {c['SQ_ACTIVE_INST_VALU']*4/ max(c['SQ_BUSY_CYCLES'],1)/4:5.2f} lanes {c['SQ_THREAD_CYCLES_VALU']/max(c['SQ_ACTIVE_INST_VALU'],1)/4:5.1f}"
          f" insts_valu {c['SQ_INSTS_VALU']/1e6:8.1f}M vmem_rd {c['SQ_INSTS_VMEM_RD']/1e6:7.2f}M wr {c['SQ_INSTS_VMEM_WR']/1e6:6.2f}M lds {c['SQ_INSTS_LDS']/1e6:7.2f}M ta_busy {c['TA_BUSY_avr']/max(g,1):5.2f}"
          f" lds_active/busy {c['SQ_ACTIVE_INST_LDS']/max(c['SQ_BUSY_CYCLES'],1):5.2f} valu_active/busy {c['SQ_ACTIVE_INST_VALU']/max(c['SQ_BUSY_CYCLES'],1):5.2f} wavecyc/busy {c['SQ_WAVE_CYCLES']/max(c['SQ_BUSY_CYCLES'],1):5.1f}")
PY
