#!/bin/bash
# round 5, first visit: the new run partition (tests + batch sweep against the round-4 partition) and the co-residency kill-test
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5a; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1 || { echo "smoke failed"; tail -5 $OUT/smoke.txt; exit 1; }
tail -1 $OUT/smoke.txt
timeout 600 python -m pytest tests/test_gpu_stft_gl.py tests/test_gpu_full_size.py -x -q -s > $OUT/pytest_gl.log 2>&1; echo "pytest rc=$?"; grep -E " passed| failed" $OUT/pytest_gl.log | tail -2
timeout 300 python tools/probe_batch_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/sweep_new.txt; cat $OUT/sweep_new.txt
RFX_LIB_PATH=$R/build_var/librfx_r4part.so timeout 300 python tools/probe_batch_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/sweep_r4.txt; cat $OUT/sweep_r4.txt
timeout 400 python tools/probe_overlap.py 2>&1 | grep -v amdgpu.ids > $OUT/overlap.txt; cat $OUT/overlap.txt
