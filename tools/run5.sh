cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
RFX_FWD_V1=1 REF=/tmp/mel_ref.pt python tools/probe_fwd2.py 2>&1 | grep -v amdgpu.ids
for r in 64 32 16 8; do echo "= run cap $r"; RFX_FWD_RUN=$r CMP=/tmp/mel_ref.pt python tools/probe_fwd2.py 2>&1 | grep -v amdgpu.ids; done
for v in "$@"; do echo "= $v"; RFX_LIB_PATH=$GRAFT_REPO_ROOT/build_var/librfx_$v.so CMP=/tmp/mel_ref.pt python tools/probe_fwd2.py 2>&1 | grep -v amdgpu.ids; done
timeout 900 python -m pytest tests/test_gpu_mel_codec.py tests/test_gpu_api_contract.py tests/test_gpu_boundary_round2.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
} 2>&1 | tee gpurun_out/run5.log
