set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -x -q) > gpurun_out/r03_all_tests.log 2>&1
tail -5 gpurun_out/r03_all_tests.log
(timeout 600 python tools/stress_table_flip.py) > gpurun_out/r03_tf_stress.log 2>&1
tail -3 gpurun_out/r03_tf_stress.log
for R in 1024 2048; do
for f in smol_amd/exp/libsmolmc_wl_*.so; do
  t=$(SMOLMC_LIB=$PWD/$f python tools/bench_configs.py --config 4 --replicas $R --mc 20000 2>&1 | grep -v "^\[" | tail -2 | tr '\n' ' ' | cut -c1-330)
  echo "wl $R $(basename $f) $t"
done; done > gpurun_out/r03_wl_exp2.txt
cat gpurun_out/r03_wl_exp2.txt
python tools/equil_sweep.py --config 5 --T 400:2000 2500:12500 --mu 0.5 --equil 400000 > gpurun_out/r03_sweep5d.jsonl 2> gpurun_out/r03_sweep5d.err
cut -c1-520 gpurun_out/r03_sweep5d.jsonl
SMOLMC_LIB=$PWD/smol_amd/exp/libsmolmc_tfphases.so python tools/bench_configs.py --config 5 --launches 1 2>&1 | grep -i "phases" | tail -6 > gpurun_out/r03_tf_phases2.txt
cat gpurun_out/r03_tf_phases2.txt
for k in 3 6 7 9; do python tools/bench_configs.py --config $k 2>/dev/null | tail -1 | cut -c1-420; done > gpurun_out/r03_gx_ab2.jsonl
cat gpurun_out/r03_gx_ab2.jsonl
