#!/bin/bash
# One GPU visit: rocprofv3 kernel stats of the bench command, then PMC passes (each counter group in its own run, with
# --kernel-trace only) for the two dominant kernels: rfx::gl_iter_kernel<2> (decode) and rfx::stft_mel2_kernel (forward).
# The InverseMelScale kernel has its own script (tools/pmc_imel.sh: 64-tile launches only).
# Summaries land in gpurun_out/prof/; copy the ones to be judged into profiles/.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"  # (the batch sweep would mix other launch shapes into the per-kernel averages)
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BENCH > $OUT/stats.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$tag -o p -- $BENCH --steps 1 > $OUT/pmc_$tag.log 2>&1
done
python - <<PY
import csv, glob, json, collections
out = "$OUT"
def agg(counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(out + "/pmc_*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            a = tot[r["Kernel_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in tot.items()}
names = ["FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES",
         "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
         "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE", "GRBM_COUNT"]
tables = {c: agg(c) for c in names}
try:
    rev = open("$R/.git_rev").read().strip()   # written by tools/gpu.sh: the pushed snapshot has no .git
except Exception:
    rev = "not recorded"
import sys
sys.path.insert(0, "$R")
import bench  # kernel_source_fingerprint: comment- and whitespace-insensitive (round 6)
def src_sha(files):
    return bench.kernel_source_fingerprint(tuple(files))
SRC = {"gl_iter_pmc.json": ("rfx_gl.hip", "rfx_core.h", "rfx_frame.hip.h", "rfx_kernels.h"), "forward_pmc.json": bench.FORWARD_KERNEL_SOURCES}
for key, fname in (("gl_iter_kernel<2>", "gl_iter_pmc.json"), ("stft_mel2_kernel", "forward_pmc.json")):
    res = {}
    for k in tables["FETCH_SIZE"]:
        if key not in k: continue
        f_kb = tables["FETCH_SIZE"][k][1]; w_kb = tables["WRITE_SIZE"].get(k, (0, 0.0))[1]
        res = {"kernel": k, "git": rev, "src_sha": src_sha(SRC[fname]), "src_files": list(SRC[fname]), "batch_tiles": 64, "frames_per_tile": 512, "launches_sampled": tables["FETCH_SIZE"][k][0], "FETCH_SIZE_kb_raw": f_kb, "WRITE_SIZE_kb_raw": w_kb,
               "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
               "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE uncorrected; "
                       "each counter group collected in its own rocprofv3 --kernel-trace --pmc run of bench.py"}
        for c in names[2:]:
            if k in tables[c]: res[c + "_per_launch"] = tables[c][k][1]
        # launch duration inside the counter runs (the chip clocks differently under the profiler): from their kernel traces
        durs = []
        for f in glob.glob(out + "/pmc_GRBM_GUI_ACTIVE/*kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                if key in r.get("Kernel_Name", ""): durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        if durs: res["profiled_launch_ms"] = sum(durs) / len(durs)
    json.dump(res, open(out + "/" + fname, "w"), indent=1)
    print(fname, json.dumps(res)[:600])
PY
cp $OUT/stats/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; head -12 $OUT/kernel_stats.csv
