# universal kernel A/B (round 5): forced-universal configurations with the dictionaries in LDS / in global memory and
# with / without the shadow copies of the feature cells.   bash tools/ab_univ.sh   (on the GPU box, from the repo root)
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/univ_tests.log 2>&1; tail -5 gpurun_out/univ_tests.log
for env in "X=1" "SMOLMC_UNIV_NO_DICT=1" "SMOLMC_UNIV_NO_COPIES=1" "SMOLMC_UNIV_NO_DICT=1 SMOLMC_UNIV_NO_COPIES=1"; do
  for cfgargs in "--config 2 --mc 2000" "--config 2 --mc 2000 --replicas 2048" "--config 5" "--config 3 --mc 1000" "--config 8 --mc 1000"; do
    echo "== $env $cfgargs"; env $env SMOLMC_FORCE_UNIVERSAL=1 timeout 300 python tools/bench_configs.py $cfgargs 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['kernel'], 'ms', round(d['kernel_ms'],3), 'steps/s %.4g'%d['mc_steps_per_s'], 'flips/s %.4g'%d['flips_per_s'], 'acc', round(d['acceptance'],4))"
  done
done > gpurun_out/univ_ab.log 2>&1
cat gpurun_out/univ_ab.log
