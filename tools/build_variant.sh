#!/bin/bash
# build_var/librfx_<name>.so: the library with extra compiler flags (diagnostic / A-B builds; the product is __graft_entry__.build())
#   bash tools/build_variant.sh wgclock -DRFX_WGCLOCK
set -e
R=$(cd "$(dirname "$0")/.." && pwd); name=$1; shift
mkdir -p $R/build_var
cd $R/riffusion-hobby_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-result -fno-slp-vectorize -I $R/include "$@" *.hip -o $R/build_var/librfx_$name.so
ls -la $R/build_var/librfx_$name.so
