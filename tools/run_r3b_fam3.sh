cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RATES=48000 B=16 RFX_LIB_PATH=$GRAFT_REPO_ROOT/build_var/librfx_famtiming.so python tools/probe_generic.py 2>&1 | grep -v amdgpu.ids | grep timing | tail -12 | tee gpurun_out/r3b_fam_timing.log
