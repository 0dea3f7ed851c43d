cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mel_codec.py tests/test_gpu_full_parity.py tests/test_gpu_round3_parity.py tests/test_gpu_api_contract.py tests/test_gpu_generic_geometry.py -m gpu -x -q -s -k "inverse or imel or mel or slaney or tile or early or stop" > gpurun_out/r3b_imel_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3b_imel_pytest.log
grep -E "rel-L2|dB" gpurun_out/r3b_imel_pytest.log | head -30
for v in default imel_nouf imel_old; do
  echo "=== $v"
  if [ $v = default ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$GRAFT_REPO_ROOT/build_var/librfx_$v.so; fi
  python tools/probe_imel.py 2>&1 | grep -v amdgpu.ids | tail -3
done | tee gpurun_out/r3b_variants_imel.log
