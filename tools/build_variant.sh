#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...]  ->  build_var/librfx_NAME.so  (A/B builds for tools/run_variants.sh)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-result -fno-slp-vectorize \
  -I include "$@" riffusion-hobby_amd/csrc/*.hip -o build_var/librfx_$name.so
echo "built build_var/librfx_$name.so"
