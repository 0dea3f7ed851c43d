cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_generic_geometry.py -m gpu -x -q -s > gpurun_out/r3b_fam_pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/r3b_fam_pytest.log
grep -E "Error|error|assert" gpurun_out/r3b_fam_pytest.log | grep -v "^ *#" | head -10
RATES=48000,22050,16000 timeout 200 python tools/probe_generic.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3b_fam_probe.log
