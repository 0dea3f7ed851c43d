#!/bin/bash
# round 5, fifth visit: run lengths by dispatch order - parity with the skew on, then the sweep
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5e; mkdir -p $OUT; cd $R
export RFX_LIB_PATH=$R/build_var/librfx_abl.so
RFX_GL_SKEW=100 RFX_GL_SKEW0=200 timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_full_size.py tests/test_gpu_stft_gl.py tests/test_gpu_full_parity.py -m gpu -q -s > $OUT/pytest_skew.log 2>&1; echo "pytest (skew 100 / 200) rc=$?"
grep -E " passed| failed" $OUT/pytest_skew.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/pytest_skew.log | head
timeout 600 python tools/probe_skew.py 2>&1 | grep -v amdgpu.ids > $OUT/skew_sweep.txt; cat $OUT/skew_sweep.txt
