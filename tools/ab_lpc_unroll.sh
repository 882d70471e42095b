# A/B of build_var/*.so variants of the one-evaluation-per-lane kernels against the in-tree library, alternating (tools/build_variant.sh)
#   tools/ab_lpc_unroll.sh "VARIANTS" "SIZES" "ORDERS"
vars=${1:-lpc2}; sizes=${2:-8192,16384,32768,65536,1048576}; orders=${3:-5 3}
export CARMA_TUNE_LPC_MIN=4096
for pq in "5 3" "7 6" "3 2"; do
case " $orders " in *" ${pq% *} "*) ;; *) continue;; esac
for rep in 1 2; do
for w in $vars main; do
  if [ $w = main ]; then unset CARMA_LIB_PATH; else export CARMA_LIB_PATH=$PWD/build_var/$w.so; fi
  echo "== $w ($pq)"; LANE_PROBE_B=$sizes python tools/lane_probe.py $pq 2>&1 | grep "B=" | cut -c1-100
done; done; done
