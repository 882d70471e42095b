#!/bin/bash
# A/B builds of the same C ABI: tools/build_variant.sh NAME [extra hipcc flags]  ->  build_var/NAME.so
# (select it at run time with CARMA_LIB_PATH=$PWD/build_var/NAME.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_var/$name
SRC=carma_pack_amd/csrc
for f in carma_kernels carma_capi carma_pt carma_pt_host carma_shard carma_mle carma_post carma_pt_lane; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function "$@" -Iinclude -I$SRC -c $SRC/$f.hip -o build_var/$name/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build_var/$name/*.o -o build_var/$name.so -ldl
echo built build_var/$name.so
