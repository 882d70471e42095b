#!/bin/bash
# Diagnostic build with per-segment cycle stamps in the filter loop (see tools/diag_stamps.py).
set -e
cd "$(dirname "$0")/.."
mkdir -p build_diag
SRC=carma_pack_amd/csrc
for f in carma_kernels carma_capi carma_pt carma_pt_host carma_shard carma_mle carma_post carma_pt_lane; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC ${DIAG_FLAGS:--DCARMA_STAMPS} -Iinclude -I$SRC -c $SRC/$f.hip -o build_diag/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build_diag/*.o -o build_diag/libcarma_mi355_diag.so -ldl
echo built build_diag/libcarma_mi355_diag.so
