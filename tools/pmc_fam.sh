#!/bin/bash
# SQ / HBM counters of the row-family Griffin-Lim kernel at 48 kHz (fam_gl_kernel<1, 24, 20, 40, 0>: the iterating mode, 64 tiles per launch): separate
# --pmc passes with --kernel-trace only, every rocprofv3 run under its own timeout.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_fam; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" ; do
  i=$((i+1))
  B=64 RATES=48000 timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p -- python $R/tools/probe_fam.py > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: [0, 0.0])
kernel = ""
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "fam_gl_kernel<1" not in r.get("Kernel_Name", ""): continue
        kernel = r["Kernel_Name"]
        a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
res = {"kernel": kernel, "batch_tiles": 64, "frames_per_tile": 512, "sample_rate": 48000,
       "note": "per-launch averages; FETCH_SIZE counts 64-byte units for 128-byte requests on gfx950 (MI355X_MICROARCH.md): hbm_bytes_per_launch = (2 FETCH + WRITE) KB; "
               "each counter group in its own rocprofv3 --kernel-trace --pmc run of tools/probe_fam.py"}
for k, (n, v) in sorted(agg.items()):
    res[k + "_per_launch"] = v / n
    res.setdefault("launches_sampled", n)
if "FETCH_SIZE_per_launch" in res:
    res["hbm_bytes_per_launch"] = (2.0 * res["FETCH_SIZE_per_launch"] + res.get("WRITE_SIZE_per_launch", 0.0)) * 1024.0
try:
    res["git"] = open("$R/.git_rev").read().strip()
except Exception:
    pass
json.dump(res, open("$OUT/fam_pmc.json", "w"), indent=1)
print(json.dumps(res)[:900])
PY
