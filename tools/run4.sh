cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
echo "== forward: table form (RFX_FWD_V1) then product form"
RFX_FWD_V1=1 REF=/tmp/mel_ref.pt python tools/probe_fwd2.py 2>&1 | grep -v amdgpu.ids
CMP=/tmp/mel_ref.pt python tools/probe_fwd2.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do echo "= $v"; RFX_LIB_PATH=$GRAFT_REPO_ROOT/build_var/librfx_$v.so CMP=/tmp/mel_ref.pt python tools/probe_fwd2.py 2>&1 | grep -v amdgpu.ids; done
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_mel_codec.py tests/test_gpu_api_contract.py tests/test_gpu_boundary_round2.py tests/test_gpu_generic_geometry.py -x -q 2>&1 | tail -5
} 2>&1 | tee gpurun_out/run4.log
