set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/r03_all_tests2.log 2>&1
tail -5 gpurun_out/r03_all_tests2.log | cut -c1-300
for R in 1024 2048; do python tools/bench_configs.py --config 4 --replicas $R --mc 20000 2>/dev/null | tail -1 | cut -c1-400; done > gpurun_out/r03_wl_final.jsonl
cat gpurun_out/r03_wl_final.jsonl
