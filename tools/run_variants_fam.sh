# usage: run_variants_fam.sh name...   (build_var/librfx_<name>.so built by tools/build_variants.sh; tools/probe_fam.py with each, the shipped library first and last)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in default "$@" default; do
  if [ $v = default ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$GRAFT_REPO_ROOT/build_var/librfx_$v.so; fi
  TAG=$v RATES=${RATES:-48000,16000} timeout 120 python tools/probe_fam.py 2>&1 | grep -v amdgpu.ids | tail -1
done | tee gpurun_out/variants_fam.log
