#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_fwd48; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RATE=48000 timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fwd -- python $R/tools/probe_fwd_rate.py > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-170
