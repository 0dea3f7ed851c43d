#!/bin/bash
# round 5: the diagnostic probes behind DESIGN 4.1's "power-bound" and "dispatch order" paragraphs, in one visit
#   (here)   bash tools/build_variants.sh wgclock:"-DRFX_WGCLOCK" abl:""
#   (GPU)    bash tools/gpu.sh --timeout 900 -- 'bash tools/gpu_r5_probes.sh'
# Results land in gpurun_out/r5p/; what is to be judged is copied into profiles/ (r05_power_clock_probe.txt, r05_wgclock_dispatch_order.txt).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5p; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1 || { echo "smoke failed"; tail -5 $OUT/smoke.txt; exit 1; }
# clock and socket power under each kernel of the path (the product library)
timeout 200 python tools/probe_power.py 2>&1 | grep -v amdgpu.ids > $OUT/power.txt; cut -c1-300 $OUT/power.txt
# per-workgroup clocks of every Griffin-Lim launch, equal runs against runs skewed by dispatch order
for skew in 0 100; do
  echo "== RFX_GL_SKEW=$skew" >> $OUT/wgclock_skew.txt
  RFX_LIB_PATH=$R/build_var/librfx_wgclock.so RFX_GL_SKEW=$skew timeout 200 python tools/probe_wgclock.py 2>&1 | grep -v amdgpu.ids >> $OUT/wgclock_skew.txt
done
cut -c1-300 $OUT/wgclock_skew.txt
# parity with the skew on, then the sweep
RFX_LIB_PATH=$R/build_var/librfx_abl.so RFX_GL_SKEW=100 RFX_GL_SKEW0=200 timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_full_size.py tests/test_gpu_stft_gl.py -m gpu -q > $OUT/pytest_skew.log 2>&1; tail -2 $OUT/pytest_skew.log
RFX_LIB_PATH=$R/build_var/librfx_abl.so timeout 600 python tools/probe_skew.py 2>&1 | grep -v amdgpu.ids > $OUT/skew_sweep.txt; cat $OUT/skew_sweep.txt
