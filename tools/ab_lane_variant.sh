#!/bin/bash
# A/B of the lane kernels between the in-tree library and build_var/$1.so (tools/build_variant.sh), alternating, three rounds:
#   tools/ab_lane_variant.sh NAME     (on the GPU box)
v=${1:?variant name}
for rep in 1 2 3; do
for w in $v main; do
  if [ $w = main ]; then unset CARMA_LIB_PATH; else export CARMA_LIB_PATH=$PWD/build_var/$w.so; fi
  echo "== $w"; LANE_PROBE_B=${LANE_PROBE_B:-16384,65536,262144} python tools/lane_probe.py 2>&1 | grep "B=" | cut -c1-80
done; done
