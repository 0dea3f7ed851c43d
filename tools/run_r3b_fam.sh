cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_generic_geometry.py -m gpu -x -q -s -k "row_family or griffinlim or agrees" > gpurun_out/r3b_fam_pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/r3b_fam_pytest.log
grep -E "Error|error|assert" gpurun_out/r3b_fam_pytest.log | grep -v "^ *#" | head
RATES=48000,32000,24000,16000,8000 python tools/probe_generic.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3b_fam_probe.log
cd /tmp && export TMPDIR=/tmp
RATES=48000 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fam -o fam -- python $GRAFT_REPO_ROOT/tools/probe_generic.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_fam -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
