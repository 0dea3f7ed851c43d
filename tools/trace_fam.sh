#!/bin/bash
# rocprofv3 kernel stats of the row-family Griffin-Lim at 48 kHz (tools/probe_fam.py: 64 tiles, Griffin-Lim 32, five calls)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_fam; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=64 RATES=48000 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fam -- python $R/tools/probe_fam.py > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
