set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/bench_r01_final.json 2> gpurun_out/bench_r01_final.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_final_$n -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/pmc_final_$n.log 2>&1
done
cd $R
python tools/rocpd_summary.py gpurun_out/prof_final gpurun_out/pmc_final_* > gpurun_out/final_summary.txt 2>&1
tail -5 gpurun_out/bench_r01_final.json
for c in 1 3 4 5 6 7; do python tools/bench_configs.py --config $c > gpurun_out/config${c}_final.json 2> gpurun_out/config${c}_final.err; done
cat gpurun_out/config*_final.json
