# timing-only A/B builds of the wave pipeline: the mean wave / the producers / both stop working after the first chunks (their
# barriers stay), so that the covariance wave runs without their LDS traffic and issue load -- how much of its 197 cycles per
# pass is interference (its pass alone: 162-167 cycles, tools/ubench/ub9_pass.hip)
for rep in 1 2 3; do
for v in ${VARIANTS:-nomean noprod noboth main}; do
  if [ $v = main ]; then unset CARMA_LIB_PATH; else export CARMA_LIB_PATH=$PWD/build_var/$v.so; fi
  echo -n "$v: "; timeout 300 python bench.py --no-cpu --no-pipelined --no-mcmc --no-throughput --no-ladder --steps 3000 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1000,3), round(d['roofline']['kernel_avg_us'],3))"
done; done
