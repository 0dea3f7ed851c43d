#!/bin/bash
# SQ counters of the InverseMelScale kernel (separate --pmc passes, kernel trace only)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_imel; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p1 -o p -- python $R/tools/probe_imel.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -o p -- python $R/tools/probe_imel.py > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, collections, json
res = {}
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "imel_group_kernel" not in r.get("Kernel_Name", ""): continue
        v = float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES" and v < 1000: continue   # the early-stop relaunch exits immediately
        a = agg[r["Counter_Name"]]; a[0] += 1; a[1] = max(a[1], v)
    for c, (n, v) in agg.items(): res[c] = v
json.dump(res, open("$OUT/imel_pmc.json", "w"), indent=1)
print(json.dumps(res))
PY
