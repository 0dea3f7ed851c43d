#!/bin/bash
# Builds experiment variants of librfx.so into build_var/ (git-ignored; travels to the GPU box).  (tools/build_variant_fast.sh does the same
# for ONE variant and recompiles only the translation units the flags concern: 35 s instead of minutes.)
#   tools/build_variants.sh name1:"-DFLAG ..." name2:"..."     (source dir override: SRC=/path/to/csrc)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${SRC:-$ROOT/riffusion-hobby_amd/csrc}
mkdir -p $ROOT/build_var
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  ( cd $SRC && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-result -fno-slp-vectorize -DRFX_ABLATION \
      -I $ROOT/include $flags *.hip -o $ROOT/build_var/librfx_$name.so ) &
done
wait
ls -la $ROOT/build_var
