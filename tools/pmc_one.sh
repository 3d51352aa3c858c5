cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc1 -o vm -- python3 tools/build_only.py lost_empire_4k sah-wide 2 > gpurun_out/pmc1.log 2>&1
python3 - <<'PY'
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("gpurun_out/pmc1/vm_counter_collection.csv")):
    agg[r["Kernel_Name"].split("(")[0][:30]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in agg.items():
    if "sah_group" in k or "wide_level" in k or "refit" in k:
        w = max(c["SQ_WAVE_CYCLES"], 1)
        print(k, {n: round(v / w, 3) for n, v in c.items()}, "wave_cycles", c["SQ_WAVE_CYCLES"])
PY
