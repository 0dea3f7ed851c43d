#!/bin/bash
# One GPU visit: bench line, rocprofv3 kernel stats of the same command, PMC passes for HBM traffic.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
python - <<PY
import csv, glob, json, collections
out = "$OUT"
def agg(pattern, counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            a = tot[r["Kernel_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in tot.items()}
fetch = agg(out + "/pmc_fetch/*counter_collection.csv", "FETCH_SIZE")
write = agg(out + "/pmc_write/*counter_collection.csv", "WRITE_SIZE")
sq = {c: agg(out + "/pmc_sq/*counter_collection.csv", c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")}
res = {}
for k in fetch:
    if "gl_iter_kernel<2>" in k:
        f_kb = fetch[k][1]; w_kb = write.get(k, (0, 0.0))[1]
        # FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1 KB per rocprof docs; gfx950 FETCH_SIZE under-counts
        # wide coalesced reads by exactly 2x (MI355X_MICROARCH.md, HBM section) -> doubled here
        res = {"kernel": k, "launches_sampled": fetch[k][0], "FETCH_SIZE_kb_raw": f_kb, "WRITE_SIZE_kb_raw": w_kb,
               "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
               "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE uncorrected"}
        for c, table in sq.items():
            if k in table: res[c + "_per_launch"] = table[k][1]
json.dump(res, open(out + "/gl_iter_pmc.json", "w"), indent=1)
print(json.dumps(res))
PY
ls $OUT/stats | head
