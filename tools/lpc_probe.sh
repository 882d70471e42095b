#!/bin/bash
# one evaluation per lane with producer waves (k_logdens_carma_lpc<P,3>: 1 consumer + 3 producers per 64 evaluations) /
# the plain lane kernel / the lane-group kernels / the library's own dispatch, per batch size.
# (The 1 + 1 producer form of round 3 is not in the library any more: profiles/r03/lpc_orders_v1.txt records it.)
# LPC_PROBE_WHICH bit 0: producer waves everywhere, 1: plain lane kernel, 2: lane-group kernels, 3: default dispatch
export LANE_PROBE_B=${LANE_PROBE_B:-6144,8192,12288,16384,24576,32768,49152,65536,98304,131072}
ARGS="$@"
W=${LPC_PROBE_WHICH:-15}
[ $((W & 1)) -ne 0 ] && CARMA_TUNE_LPC_MIN=0 CARMA_TUNE_LPC_MAX=999999999 python tools/lane_probe.py $ARGS 2>&1 | grep -v amdgpu
[ $((W & 2)) -ne 0 ] && CARMA_TUNE_LPC_MAX=0 CARMA_TUNE_LANE_MIN=0 python tools/lane_probe.py $ARGS 2>&1 | grep -v amdgpu
[ $((W & 4)) -ne 0 ] && CARMA_TUNE_LPC_MAX=0 CARMA_TUNE_LANE_MIN=999999999 python tools/lane_probe.py $ARGS 2>&1 | grep -v amdgpu
[ $((W & 8)) -ne 0 ] && python tools/lane_probe.py $ARGS 2>&1 | grep -v amdgpu
true
