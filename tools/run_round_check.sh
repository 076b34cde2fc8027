# Round-end check on the GPU box (from the repo root): the GPU test tier, the headline profile (bench line,
# rocprofv3 kernel stats, PMC passes -> per-kernel-stamped constants) and a time-boxed differential campaign.
#   bash tools/run_round_check.sh r05b [campaign minutes]
# Outputs under gpurun_out/ (scratch); copy what should be judged into profiles/.
tag=${1:-r05b}
mins=${2:-2}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gputests.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/${tag}_gputests.txt
tail -3 gpurun_out/${tag}_gputests.txt
timeout 900 bash tools/run_profile.sh ${tag} > gpurun_out/${tag}_run_profile.log 2>&1
echo "run_profile rc=$?"
cd $R
timeout $((mins * 60 + 240)) python tests/fuzz_campaign.py --profile fast --minutes ${mins} --cases 100000 --seed 91000 --out gpurun_out/${tag}_fuzz_fast.jsonl > gpurun_out/${tag}_fuzz_fast.txt 2>&1
tail -4 gpurun_out/${tag}_fuzz_fast.txt
timeout $((mins * 60 + 240)) python tests/fuzz_campaign.py --profile any --minutes ${mins} --cases 100000 --seed 92000 --out gpurun_out/${tag}_fuzz_any.jsonl > gpurun_out/${tag}_fuzz_any.txt 2>&1
tail -4 gpurun_out/${tag}_fuzz_any.txt
