for rep in 1 2 3; do
for v in ${VARIANTS:-pre_phase phase_only sym_v1 main}; do
  if [ $v = main ]; then unset CARMA_LIB_PATH; else export CARMA_LIB_PATH=$PWD/build_var/$v.so; fi
  echo -n "$v: "; timeout 200 python bench.py --no-cpu --no-pipelined --no-mcmc --no-throughput --no-ladder --steps 3000 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1000,3), round(d['roofline']['kernel_avg_us'],3))"
done; done
