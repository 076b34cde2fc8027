set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q -k "table or mson or fullsize or fuzz or moca") > gpurun_out/r03_tf_tests.log 2>&1
tail -5 gpurun_out/r03_tf_tests.log
(timeout 600 python tools/stress_table_flip.py) > gpurun_out/r03_tf_stress.log 2>&1
tail -3 gpurun_out/r03_tf_stress.log
for f in smol_amd/exp/libsmolmc_wl_*.so; do
  t=$(SMOLMC_LIB=$PWD/$f python tools/bench_configs.py --config 4 --replicas 1024 --mc 20000 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],4), d['acceptance'])")
  echo "wl $(basename $f) $t"
done > gpurun_out/r03_wl_exp.txt
t=$(SMOLMC_NO_SOLO=1 python tools/bench_configs.py --config 2 --replicas 1024 --mc 20000 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],4), d['acceptance'], d['kernel'])")
echo "metropolis no-solo 1024 $t" >> gpurun_out/r03_wl_exp.txt
cat gpurun_out/r03_wl_exp.txt
python tools/equil_sweep.py --config 5 --T 400:2000 2500:12500 --mu 0.5 --equil 400000 > gpurun_out/r03_sweep5c.jsonl 2> gpurun_out/r03_sweep5c.err
cut -c1-520 gpurun_out/r03_sweep5c.jsonl
SMOLMC_LIB=$PWD/smol_amd/exp/libsmolmc_tfphases.so python tools/bench_configs.py --config 5 --launches 1 2>&1 | grep -i "phases" | tail -2 > gpurun_out/r03_tf_phases.txt
cat gpurun_out/r03_tf_phases.txt
