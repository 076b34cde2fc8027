set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SMOLMC_LIB=$PWD/smol_amd/exp/libsmolmc_wlphases.so python tools/bench_configs.py --config 4 --replicas 1024 --mc 20000 --launches 1 2>&1 | grep -i "phases" | tail -2 > gpurun_out/r03_wl_phases.txt
cat gpurun_out/r03_wl_phases.txt
(timeout 900 python -m pytest tests -m gpu -x -q -k "ewald or field or mson or fullsize or fuzz or table or bias or moca") > gpurun_out/r03_gx_tests.log 2>&1
tail -5 gpurun_out/r03_gx_tests.log
for k in 3 5 6 7; do
  python tools/bench_configs.py --config $k 2>/dev/null | tail -1 | cut -c1-600
  SMOLMC_NO_EWALD_GX=1 python tools/bench_configs.py --config $k 2>/dev/null | tail -1 | cut -c1-600
done > gpurun_out/r03_gx_ab.jsonl
cat gpurun_out/r03_gx_ab.jsonl
python tools/bench_mson.py --dim 12 --walkers 4096 --mc 1000 2>/dev/null > gpurun_out/r03_mson12_gx.jsonl
SMOLMC_NO_EWALD_GX=1 python tools/bench_mson.py --dim 12 --walkers 4096 --mc 1000 2>/dev/null > gpurun_out/r03_mson12_nogx.jsonl
cut -c1-400 gpurun_out/r03_mson12_gx.jsonl gpurun_out/r03_mson12_nogx.jsonl
