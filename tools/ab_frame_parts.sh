# the co-rotating producer-wave kernel, its consumer alone (lf_noprod) and its producers alone (lf_nocons): tools/build_variant.sh
export LANE_PROBE_B=${LANE_PROBE_B:-8192,16384,32768}
for post in 1.0 0.5; do
export LANE_PROBE_POST=$post
for w in main lf_noprod lf_nocons; do
  if [ $w = main ]; then unset CARMA_LIB_PATH; else export CARMA_LIB_PATH=$PWD/build_var/$w.so; fi
  echo "== $w (posterior-like fraction $post)"; CARMA_TUNE_LPC_MIN=3072 python tools/lane_probe.py 2>&1 | grep "B=" | cut -c1-90
done; done
