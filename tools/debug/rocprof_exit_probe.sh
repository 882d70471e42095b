# Which process exits with a fault under rocprofv3 (rc 139 AFTER its outputs are written)?  Any that made a cooperative launch -- the row
# sampler k_pt_row (hipLaunchCooperativeKernel); log-density launches, the lane sampler, a plain torch script: rc 0.  Without the
# profiler every run exits 0.  (round 5; the profile passes of tools/profile_round.sh include the sampler leg, hence rc 139 there)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/samp.py <<'PY'
import sys, gc, numpy as np
sys.path.insert(0, sys.argv[1])
import carma_pack_amd as cpa
g = np.load(sys.argv[1] + "/tests/golden/carma53_readme.npz")
ctx = cpa.Context(g["t"], g["y"], g["yerr"], 5, 3)
ctx.pt_create(16, int(sys.argv[2]), adapt_iters=10 ** 9, seed=3); ctx.pt_start(None); ctx.pt_iterate(50)
print("sampler ok", ctx.pt_kernel())
if sys.argv[3] == "del":
    del ctx; gc.collect(); print("freed")
PY
for args in "64 keep" "64 del" "2 keep" "1024 keep"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/s -o a -- python3 /tmp/samp.py $R $args > /tmp/s.log 2>&1; echo "sampler [$args] rc=$? $(grep -c 'sampler ok' /tmp/s.log) $(grep 'sampler ok' /tmp/s.log)"
done
