cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 100 python tools/probe_imel.py; RFX_IMEL_NO_PAIR=1 timeout 100 python tools/probe_imel.py; timeout 100 python tools/probe_imel.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3b_pair_probe.log
timeout 600 python -m pytest tests/test_gpu_full_parity.py tests/test_gpu_round3_parity.py tests/test_gpu_mel_codec.py tests/test_gpu_api_contract.py tests/test_gpu_boundary_round2.py -m gpu -x -q -s -k "mono_tile or stereo_tile or slaney or inside or inverse or early or stop or imel or coupl" > gpurun_out/r3b_pair_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r3b_pair_pytest.log | tail -1; grep -E "rel-L2" gpurun_out/r3b_pair_pytest.log | cut -c1-200 | head
