#!/bin/bash
# SQ counters of the InverseMelScale kernel (separate --pmc passes, kernel trace only), averaged over the 64-tile launches ONLY:
# the fix-up launch that re-runs early-stopped clips has the same grid but exits at once (its SQ_INSTS_VALU is ~1e5), and
# launches on fewer tiles have a smaller grid - both are filtered out by grid size and instruction count.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_imel; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" ; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p -- env FORMS=${FORMS:-auto} REPS=8 python $R/tools/probe_imel.py > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json, subprocess
B, T, THREADS = 64, 512, 256
rows = collections.defaultdict(dict)   # (file, dispatch) -> {counter: value}
meta = {}
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "imel_group_kernel" not in r.get("Kernel_Name", "") and "imel_wave_kernel" not in r.get("Kernel_Name", ""): continue
        key = (f, r["Dispatch_Id"])
        rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
        meta[key] = (r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0))
agg = collections.defaultdict(lambda: [0, 0.0])
kernel = ""
for key, c in rows.items():
    name, grid = meta[key]
    if grid and grid not in (B * T * THREADS, B * T * 64): continue                  # not a 64-tile launch (group kernels: 256 threads per frame, wave kernel: 64)
    if c.get("SQ_INSTS_VALU", 1e12) < 1e7: continue                                 # the fix-up launch that exits immediately
    kernel = name
    for k, v in c.items():
        agg[k][0] += 1; agg[k][1] += v
dur = []
for f in sorted(glob.glob("$OUT/p*/*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if r.get("Kernel_Name", "") == kernel and int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) in (B * T * THREADS, B * T * 64):
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
dur = [d for d in dur if d > 1.0]  # (the fix-up launch exits at once)
res = {"kernel": kernel, "duration_ms_under_pmc": sum(dur) / max(1, len(dur)), "batch_tiles": B, "frames_per_tile": T,
       "note": "averages over the 64-tile launches only (grid = 64*512 workgroups, fix-up launches excluded); each counter group in its own rocprofv3 --kernel-trace --pmc run of tools/probe_imel.py"}
for k, (n, v) in sorted(agg.items()):
    res[k + "_per_launch"] = v / n
    res.setdefault("launches_sampled", n)
try:
    res["git"] = open("$R/.git_rev").read().strip()
except Exception:
    pass
json.dump(res, open("$OUT/imel_pmc.json", "w"), indent=1)
print(json.dumps(res))
PY
