cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_generic_geometry.py -m gpu -x -q -s -k "row_family or griffinlim or agrees" > gpurun_out/r3b_fam_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3b_fam_pytest.log
grep -E "dB|Error|error|assert" gpurun_out/r3b_fam_pytest.log | grep -v "^ *#" | head -60
RATES=48000,32000,16000 python tools/probe_generic.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3b_fam_probe.log
