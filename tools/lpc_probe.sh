#!/bin/bash
# one evaluation per lane with producer waves (1 + 3 and 1 + 1 per 64 evaluations) / without / the lane-group kernels, per batch size
export LANE_PROBE_B=${LANE_PROBE_B:-6144,8192,12288,16384,24576,32768,49152,65536,98304,131072}
ARGS="$@"
W=${LPC_PROBE_WHICH:-31}
[ $((W & 1)) -ne 0 ] && CARMA_TUNE_LPC3_MIN=0 CARMA_TUNE_LPC1_MIN=999999999 CARMA_TUNE_LPC_MAX=999999999 python tools/lane_probe.py $ARGS 2>&1 | grep -v amdgpu
[ $((W & 2)) -ne 0 ] && CARMA_TUNE_LPC3_MIN=0 CARMA_TUNE_LPC1_MIN=0 CARMA_TUNE_LPC_MAX=999999999 python tools/lane_probe.py $ARGS 2>&1 | grep -v amdgpu
[ $((W & 4)) -ne 0 ] && CARMA_TUNE_LPC_MAX=0 CARMA_TUNE_LANE_MIN=0 python tools/lane_probe.py $ARGS 2>&1 | grep -v amdgpu
[ $((W & 8)) -ne 0 ] && CARMA_TUNE_LPC_MAX=0 CARMA_TUNE_LANE_MIN=999999999 python tools/lane_probe.py $ARGS 2>&1 | grep -v amdgpu
[ $((W & 16)) -ne 0 ] && python tools/lane_probe.py $ARGS 2>&1 | grep -v amdgpu
true
