#!/bin/bash
mkdir -p gpurun_out
for v in "$@"; do
  echo "=== $v"
  RFX_LIB_PATH=$GRAFT_REPO_ROOT/build_var/librfx_$v.so python tools/probe_forward.py 2>&1 | grep -v amdgpu.ids | tail -1
done | tee gpurun_out/variants_fwd.log
