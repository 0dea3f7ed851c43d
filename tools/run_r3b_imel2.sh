cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in default prio1 prio2 cls1prio1 default; do
  echo "=== $v"
  if [ $v = default ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$GRAFT_REPO_ROOT/build_var/librfx_$v.so; fi
  python tools/probe_imel.py 2>&1 | grep -v amdgpu.ids | tail -1
done | tee gpurun_out/r3b_variants_imel2.log
unset RFX_LIB_PATH
timeout 600 python -m pytest tests/test_gpu_full_parity.py tests/test_gpu_round3_parity.py -m gpu -x -q -s -k "mono_tile or slaney or inside" 2>&1 | grep -E "rel-L2|passed|failed" | cut -c1-220
