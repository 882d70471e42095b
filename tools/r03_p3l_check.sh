#!/bin/bash
# ON THE GPU BOX: parity tests that exercise the wave pipeline's chunk handling, then its timing
mkdir -p gpurun_out/$1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sampler.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/$1/parity.txt
cat gpurun_out/$1/parity.txt
python tools/nscale_probe.py > gpurun_out/$1/nscale.txt 2>&1; grep -v amdgpu gpurun_out/$1/nscale.txt | tail -12
python tools/midrange_probe.py 2>&1 | grep "B=" > gpurun_out/$1/midrange.txt; cat gpurun_out/$1/midrange.txt
python tools/mcmc_bigR_probe.py 2>&1 | grep "R=" > gpurun_out/$1/bigR.txt; cat gpurun_out/$1/bigR.txt
python bench.py --no-cpu --no-ladder --steps 2000 > gpurun_out/$1/bench.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/$1/bench.json").read().strip().splitlines()[-1])
print("bench: value %.4g evals/s, kernel_avg_us %.2f, mcmc %.0f it/s, tput %.4g" % (d["value"], d["roofline"]["kernel_avg_us"], d["mcmc"]["iters_per_s"], d["throughput"]["evals_per_s"]))
PY
