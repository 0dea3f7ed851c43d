cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
RATES=${RATES:-48000,22050} python tools/probe_generic.py 2>&1 | grep -v amdgpu.ids
echo "= RFX_GEN_PAD=0"; RFX_GEN_PAD=0 RATES=48000 python tools/probe_generic.py 2>&1 | grep -v amdgpu.ids
echo "= timing build"; RFX_LIB_PATH=$GRAFT_REPO_ROOT/build_var/librfx_gtim.so RATES=48000 python tools/probe_generic.py 2>&1 | grep -v amdgpu | tail -3
timeout 900 python -m pytest tests/test_gpu_generic_geometry.py -x -q -s 2>&1 | grep -E "n_iter=32|passed|failed|Error|assert" | tail -12
} 2>&1 | tee gpurun_out/run6.log
