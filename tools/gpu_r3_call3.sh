set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q -k "wang or wl or fullsize or fuzz or sample_rows or moca or ewald or field or mson") > gpurun_out/r03_wl_tests2.log 2>&1
tail -5 gpurun_out/r03_wl_tests2.log
for rep in 1 2; do
for R in 1024 2048; do
  python tools/bench_configs.py --config 4 --replicas $R --mc 20000 2>/dev/null | tail -1 | cut -c1-400
done; done > gpurun_out/r03_wl_ab2.jsonl
SMOLMC_WL_V2=1 python tools/bench_configs.py --config 4 --replicas 1024 --mc 20000 2>/dev/null | tail -1 | cut -c1-400 >> gpurun_out/r03_wl_ab2.jsonl
python tools/bench_configs.py --config 2 --replicas 1024 --mc 20000 2>/dev/null | tail -1 | cut -c1-400 >> gpurun_out/r03_wl_ab2.jsonl
cat gpurun_out/r03_wl_ab2.jsonl
