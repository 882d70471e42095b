for rep in 1 2 3; do
for v in notab main; do
  if [ $v = main ]; then unset CARMA_LIB_PATH; else export CARMA_LIB_PATH=$PWD/build_var/$v.so; fi
  echo "== $v"; LANE_PROBE_B=12288,16384,32768,65536,262144 python tools/lane_probe.py 2>&1 | grep "B=" | cut -c1-80
done; done
