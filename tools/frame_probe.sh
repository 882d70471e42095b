# co-rotating producer-wave kernel (carma_lane_frame.h) against the rotating one, producer waves from 3073 evaluations
export LANE_PROBE_B=${LANE_PROBE_B:-4096,6144,8192,12288,16384,24576,32768}
export CARMA_TUNE_LPC_MIN=3072
for pq in "5 3" "7 6" "3 2"; do
  echo "== frame kernel"; python tools/lane_probe.py $pq | cut -c1-150
  echo "== rotating kernel"; CARMA_TUNE_LANE_FRAME=0 python tools/lane_probe.py $pq | cut -c1-150
done
