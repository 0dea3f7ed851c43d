#!/bin/bash
# MFMA counters of the forward path (mel_gemm_kernel): separate --pmc pass, kernel trace only.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_fwd; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o p -- python $R/tools/probe_forward.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p2 -o p -- python $R/tools/probe_forward.py > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, collections, json
res = {}
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        name = "mel_gemm" if "mel_gemm_kernel" in k else ("stft" if "stft_kernel" in k else None)
        if not name: continue
        a = agg[(name, r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for (name, c), (n, v) in agg.items(): res.setdefault(name, {})[c] = v / n
json.dump(res, open("$OUT/forward_pmc.json", "w"), indent=1)
print(json.dumps(res))
PY
