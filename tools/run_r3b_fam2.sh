cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( RATES=44100 python tools/probe_generic.py; GLFORM=frames RATES=44100 python tools/probe_generic.py; RFX_FORCE_GENERIC=1 RATES=44100 python tools/probe_generic.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3b_fam_probe2.log
