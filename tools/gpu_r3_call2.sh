set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q -k "wang or wl or fullsize or fuzz or sample_rows or moca or WangLandau or device_plumbing") > gpurun_out/r03_wl_tests.log 2>&1
tail -5 gpurun_out/r03_wl_tests.log
for rep in 1 2; do
for R in 1024 2048; do
  python tools/bench_configs.py --config 4 --replicas $R --mc 20000 2>/dev/null | tail -1 | cut -c1-400
  SMOLMC_WL_V2=1 python tools/bench_configs.py --config 4 --replicas $R --mc 20000 2>/dev/null | tail -1 | cut -c1-400
done; done > gpurun_out/r03_wl_ab.jsonl
cat gpurun_out/r03_wl_ab.jsonl
python tools/equil_sweep.py --config 9 --T 3000 --mu 0.5 --penalty 0.01 0.05 0.2 --equil 600000 > gpurun_out/r03_sweep9.jsonl 2> gpurun_out/r03_sweep9.err
python tools/equil_sweep.py --config 5 --T 2500:12500 3000:15000 --mu 0.5 --equil 400000 > gpurun_out/r03_sweep5b.jsonl 2> gpurun_out/r03_sweep5b.err
cat gpurun_out/r03_sweep9.jsonl gpurun_out/r03_sweep5b.jsonl | cut -c1-500
tail -3 gpurun_out/r03_sweep9.err
(time python bench.py --steps 20 --warmup 5) > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err
tail -c 3000 gpurun_out/r03_bench_a.json; tail -5 gpurun_out/r03_bench_a.err
