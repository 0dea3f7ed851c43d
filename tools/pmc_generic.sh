#!/bin/bash
# PMC passes over the generic engine (48 kHz, 16 tiles): where does a frame's time go?
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_gen; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export B=64 RATES=48000
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -o p -- python $R/tools/probe_generic.py > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gen_gl_kernel<2" in k or "gen_fold_kernel" in k:
            a = agg[k[:40]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, t in agg.items():
    print(k)
    for c, (n, v) in sorted(t.items()): print(f"   {c}: {v/n:.4g}")
PY
