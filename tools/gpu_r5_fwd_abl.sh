#!/bin/bash
# round 5: where the forward kernel's mel phase spends its time.  For the product library and every build_var/librfx_NAME.so
# given: time per 64 waveforms (tools/probe_fwd_rate.py) and, with PMC=1, the LDS counters of stft_mel2_kernel.
#   tools/gpu_r5_fwd_abl.sh NAME...   -> gpurun_out/fwd_abl.txt
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for v in product "$@"; do
  if [ $v = product ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$R/build_var/librfx_$v.so; fi
  TAG=$v RATES=44100 python tools/probe_fwd_rate.py 2>&1 | grep -v amdgpu.ids | tail -1
  if [ -n "$PMC" ]; then
    ( cd /tmp && rm -rf /tmp/pmc_$v && TAG=$v RATES=44100 timeout 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES \
        --output-format csv -d /tmp/pmc_$v -o p -- python $R/tools/probe_fwd_rate.py > /tmp/pmc_$v.log 2>&1 )
    python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("/tmp/pmc_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stft_mel2" not in r["Kernel_Name"]: continue
        a = tot[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
print("   counters per launch ($v):", {k: round(v / n) for k, (n, v) in tot.items()})
PY
  fi
done | tee $OUT/fwd_abl.txt
