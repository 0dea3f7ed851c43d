cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
echo "== forward: table form (RFX_FWD_V1) then product form"
RFX_FWD_V1=1 REF=/tmp/mel_ref.pt python tools/probe_fwd2.py 2>&1 | grep -v amdgpu.ids
CMP=/tmp/mel_ref.pt python tools/probe_fwd2.py 2>&1 | grep -v amdgpu.ids
echo "== imel variants"
for v in fpw1 imel2 imel2rl fpw1 imel2 imel2rl; do echo "= $v"; RFX_LIB_PATH=$GRAFT_REPO_ROOT/build_var/librfx_$v.so python tools/probe_imel.py 2>&1 | grep -v amdgpu.ids | tail -2; done
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_mel_codec.py tests/test_gpu_full_parity.py tests/test_gpu_round3_parity.py -x -q 2>&1 | tail -5
} 2>&1 | tee gpurun_out/run3.log
