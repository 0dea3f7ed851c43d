#!/bin/bash
# usage (on the GPU box): tools/probe_variants.sh name1 name2 ...  -> parity + timing of build_var/librfx_<name>.so
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for v in "$@"; do
  export RFX_LIB_PATH=$R/build_var/librfx_$v.so
  echo "== $v"
  timeout 300 python $R/tools/probe_parity.py 2>&1 | tail -1
  timeout 120 python $R/tools/probe_gl.py 2>&1 | tail -2
done 2>&1 | tee $R/gpurun_out/variants.log
