export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
SMOLMC_DENSE_EWALD=1 python tools/bench_configs.py --config 3 --mc 500 --launches 3 > gpurun_out/dense_a.json 2>&1
cat gpurun_out/dense_a.json
cd /tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-20)
  SMOLMC_DENSE_EWALD=1 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/dense_pmc_$n -- python $R/tools/bench_configs.py --config 3 --mc 500 --launches 2 > $R/gpurun_out/dense_pmc_$n.log 2>&1
done
cd $R
python tools/rocpd_summary.py gpurun_out/dense_pmc_* 2>&1 | grep -v "not a database\|\.log" > gpurun_out/dense_pmc.txt
cat gpurun_out/dense_pmc.txt | head -60
