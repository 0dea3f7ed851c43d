#!/bin/bash
# usage: run_variants_imel.sh name...   (build_var/librfx_<name>.so, built by tools/build_variants.sh)
mkdir -p gpurun_out
for v in "$@"; do
  echo "=== $v"
  RFX_LIB_PATH=$GRAFT_REPO_ROOT/build_var/librfx_$v.so python tools/probe_imel.py 2>&1 | grep -v amdgpu.ids | tail -3
done | tee gpurun_out/variants_imel.log
