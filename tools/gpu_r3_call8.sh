set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/r03_all_tests3.log 2>&1
grep -n "passed\|failed" gpurun_out/r03_all_tests3.log | tail -3
python tools/equil_sweep.py --config 9 --T 4000 6000 --mu 0.5 --penalty 0.05 --equil 600000 > gpurun_out/r03_sweep9b.jsonl 2> gpurun_out/r03_sweep9b.err
cut -c1-520 gpurun_out/r03_sweep9b.jsonl
