#!/bin/bash
# One experiment variant of librfx.so, recompiling only the translation units the flags concern; the other objects come from a
# cache (build_var/obj, compiled once per source state with -DRFX_ABLATION).
#   tools/build_variant_fast.sh NAME "-DFLAG ..." rfx_stft.hip [more.hip]      -> build_var/librfx_NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/riffusion-hobby_amd/csrc
OBJ=$ROOT/build_var/obj
name=$1; flags=$2; shift 2
mkdir -p $OBJ $ROOT/build_var/obj_$name
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fno-slp-vectorize -DRFX_ABLATION -I $ROOT/include"
pids=()
for f in $SRC/*.hip; do
  b=$(basename $f .hip)
  if [[ " $* " == *" $b.hip "* ]]; then
    ( cd $SRC && $CC $flags -c $b.hip -o $ROOT/build_var/obj_$name/$b.o ) & pids+=($!)
  elif [ ! -f $OBJ/$b.o ] || [ -n "$(find $SRC -newer $OBJ/$b.o \( -name '*.h' -o -name "$b.hip" \) | head -1)" ]; then
    ( cd $SRC && $CC -c $b.hip -o $OBJ/$b.o ) & pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
objs=""
for f in $SRC/*.hip; do
  b=$(basename $f .hip)
  if [ -f $ROOT/build_var/obj_$name/$b.o ]; then objs="$objs $ROOT/build_var/obj_$name/$b.o"; else objs="$objs $OBJ/$b.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $ROOT/build_var/librfx_$name.so
ls -la $ROOT/build_var/librfx_$name.so
