#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel stats + PMC passes of the bench command.
#   tools/profile_round.sh <tag>        -> gpurun_out/prof_<tag>/...; summaries via tools/summarize_prof.py <tag> <outdir>
# Counters are collected in their own runs (only --kernel-trace next to --pmc), the program itself
# follows "--" (no env/bash -c hop: the profiler's preloaded library has already initialised the GPU).
# The PMC passes keep the sampler leg (300 iterations of k_pt_row in two dispatches) and the throughput leg
# (k_logdens_carma<5,8,4>, 65 536 evaluations per launch) next to the headline kernel: one summary per kernel.
TAG=${1:-r04}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
# which build the counters below belong to (bench.py attaches them to a run of the same build only)
python3 -c "import sys, json; sys.path.insert(0, '$REPO'); from carma_pack_amd._lib import build_ids; json.dump(build_ids(), open('$OUT/ids.json', 'w'))"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- python3 $REPO/bench.py --no-cpu --no-pipelined --no-mcmc-large --no-api > $OUT/stats.log 2>&1
echo "stats rc=$?"
# the same without the cooperative sampler launch (whose process faults at exit under the profiler, after the statistics are written)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_nomcmc -o $TAG -- python3 $REPO/bench.py --no-cpu --no-pipelined --no-mcmc --no-mcmc-large --no-ladder --no-api > $OUT/stats_nomcmc.log 2>&1
echo "stats (no sampler legs) rc=$?"
# (--no-api: the API legs run the same kernels at other sizes -- their dispatches must not be summed into the sampler leg's)
PMC_ARGS="--steps 50 --warmup 5 --no-cpu --no-pipelined --no-ladder --no-mcmc-large --no-api --mcmc-iters 200"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o $TAG -- python3 $REPO/bench.py $PMC_ARGS > $OUT/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_SQ -o $TAG -- python3 $REPO/bench.py $PMC_ARGS > $OUT/pmc_SQ.log 2>&1
echo "pmc SQ rc=$?"
# instruction classes of the FP64 work (executed flops, not the reference's count) and the clock (GRBM_GUI_ACTIVE / 8 / time)
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_F64 -o $TAG -- python3 $REPO/bench.py $PMC_ARGS > $OUT/pmc_F64.log 2>&1
echo "pmc F64 rc=$?"
cd $REPO
# summaries on the box (what profiles/ gets), then drop the raw per-dispatch files if the 64-MiB return limit is near
python3 tools/summarize_prof.py $TAG gpurun_out/summary_$TAG > $OUT/summarize.log 2>&1; echo "summarize rc=$?"
find $OUT/stats_nomcmc -name "*kernel_stats.csv" -exec cp {} gpurun_out/summary_$TAG/kernel_stats_${TAG}_no_sampler_legs.csv \;
if [ "$(du -sm gpurun_out | cut -f1)" -gt 48 ]; then find $OUT -name "*kernel_trace.csv" -size +4M -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete; fi
du -sm gpurun_out
