#!/bin/bash
# usage: pmc.sh <variant> ; collects SQ / LDS / TCC counters for the GL kernel (separate passes)
V=${1:-c4}
export RFX_LIB_PATH=$GRAFT_REPO_ROOT/build_var/librfx_$V.so
export ITERS=4
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$V
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p -- python $GRAFT_REPO_ROOT/tools/probe_gl.py > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    agg=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")
        if "gl_iter_kernel<2>" not in k: continue
        a=agg[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for c,(n,v) in agg.items(): print(f"{c}: per-dispatch {v/n:.4g} (n={n})")
PY
