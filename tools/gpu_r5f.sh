#!/bin/bash
# round 5, sixth visit: is the Griffin-Lim kernel power-bound?  Per-workgroup clocks with and without the run-length skew
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5f; mkdir -p $OUT; cd $R
export RFX_LIB_PATH=$R/build_var/librfx_wgclock.so
for skew in 0 100 0 100; do
  echo "== RFX_GL_SKEW=$skew" >> $OUT/wgclock_skew.txt
  RFX_GL_SKEW=$skew WGCLOCK_DUMP=$OUT/rec_skew$skew.npy timeout 200 python tools/probe_wgclock.py 2>&1 | grep -v amdgpu.ids | head -5 >> $OUT/wgclock_skew.txt
done
cat $OUT/wgclock_skew.txt | cut -c1-400
{ rocm-smi --showpower --showclocks 2>&1 | head -40; } > $OUT/smi.txt 2>&1
