#!/bin/bash
# round 5, third visit: where the tail of a Griffin-Lim launch comes from (RFX_WGCLOCK build), the tests touched since the second visit
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5c; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1 || { echo "smoke failed"; tail -5 $OUT/smoke.txt; exit 1; }
tail -1 $OUT/smoke.txt
RFX_LIB_PATH=$R/build_var/librfx_wgclock.so timeout 300 python tools/probe_wgclock.py 2>&1 | grep -v amdgpu.ids > $OUT/wgclock.txt; cat $OUT/wgclock.txt
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_round3_entry.py tests/test_imel_wave_form.py tests/test_gpu_api_contract.py -m gpu -q -s > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?"
grep -E " passed| failed" $OUT/pytest_sel.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/pytest_sel.log | head
