# rocprofv3 passes for the non-headline configurations (tools/bench_configs.py): kernel-trace
# stats, then separate PMC passes.  Writes gpurun_out/configs_summary.txt.
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFGS=${CFGS:-"3 4 5 9"}
cd /tmp
for k in $CFGS; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/cprof_c$k -- python $R/tools/bench_configs.py --config $k > $R/gpurun_out/cprof_c$k.log 2>&1
  i=0
  for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/cpmc_c${k}_$i -- python $R/tools/bench_configs.py --config $k --launches 2 > $R/gpurun_out/cpmc_c${k}_$i.log 2>&1
  done
done
cd $R
for k in $CFGS; do
  echo "######## config $k"; tail -1 gpurun_out/cprof_c$k.log
  python tools/rocpd_summary.py gpurun_out/cprof_c$k gpurun_out/cpmc_c${k}_*
done > gpurun_out/configs_summary.txt 2>&1
