# Final check of the round-5 re-entry build on the GPU box (from the repo root): GPU test tier, the contract line, and the
# kernel stats + PMC passes of config 5 only (the one kernel family whose machine code moved).
tag=${1:-r05c}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gputests.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/${tag}_gputests.txt; tail -3 gpurun_out/${tag}_gputests.txt
timeout 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; tail -c 300 gpurun_out/${tag}_bench.json
cp profiles/pmc_constants.json gpurun_out/pmc_constants_${tag}.json
CFGS="5" timeout 400 bash tools/run_profile_configs.sh ${tag} > gpurun_out/${tag}_run_profile_configs.log 2>&1
echo "profile rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/pmc_constants_${tag}_all.json')); print(json.dumps(d.get('config5'), indent=0)[:900])"
