# usage: tools/pmc_sq.sh outdir -- the two SQ passes only
out=$1; mkdir -p $out
cmd="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $PMC_BENCH_ARGS"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" \
           "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $out/p$i -o vm -- $cmd > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
  echo "pass $i ok: $grp"
done
