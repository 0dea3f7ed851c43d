cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
REF=/tmp/mel_ref.pt python tools/probe_fwd2.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do echo "= $v"; RFX_LIB_PATH=$GRAFT_REPO_ROOT/build_var/librfx_$v.so CMP=/tmp/mel_ref.pt python tools/probe_fwd2.py 2>&1 | grep -v amdgpu.ids; done
REF=/tmp/mel_ref.pt python tools/probe_fwd2.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do echo "= $v"; RFX_LIB_PATH=$GRAFT_REPO_ROOT/build_var/librfx_$v.so CMP=/tmp/mel_ref.pt python tools/probe_fwd2.py 2>&1 | grep -v amdgpu.ids; done
} 2>&1 | tee gpurun_out/run5.log
