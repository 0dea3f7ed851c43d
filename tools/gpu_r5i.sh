R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5i; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_round5.py -k line_form tests/test_gpu_mel_codec.py::test_inverse_mel_other_parameter_sets_use_fallback_kernels tests/test_gpu_round3_parity.py -m gpu -q -s 2>&1 | grep -v amdgpu > $OUT/pytest.log; tail -15 $OUT/pytest.log | cut -c1-250
python tools/probe_imel_params.py 2>&1 | grep -v amdgpu | tee $OUT/imel_params.txt
