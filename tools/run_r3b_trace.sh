cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_fam
export R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B=64 RATES=48000 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fam -o fam -- python $R/tools/probe_fam.py > $R/gpurun_out/prof_fam/log.txt 2>&1
cd $R
f=$(find gpurun_out/prof_fam -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-220
