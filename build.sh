#!/bin/bash
# Build libcarma_mi355.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
SRC=carma_pack_amd/csrc
OUT=carma_pack_amd/libcarma_mi355.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function"
mkdir -p build
objs=""
for f in carma_kernels carma_pt carma_capi carma_pt_host carma_shard carma_mle carma_post carma_pt_lane; do
  if [ ! -f build/$f.o ] || [ -n "$(find $SRC include -newer build/$f.o -type f | head -1)" ]; then
    $HIPCC $FLAGS -c $SRC/$f.hip -o build/$f.o
  fi
  objs="$objs build/$f.o"
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT $objs -ldl
echo "built $OUT"
