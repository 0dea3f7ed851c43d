R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5d; mkdir -p $OUT; cd $R
WGCLOCK_DUMP=$OUT/wgclock_rec.npy RFX_LIB_PATH=$R/build_var/librfx_wgclock.so timeout 300 python tools/probe_wgclock.py 2>&1 | grep -v amdgpu.ids > $OUT/wgclock.txt; head -3 $OUT/wgclock.txt
