# extra SQ counter passes (stall attribution) for one tools/bench_configs.py configuration:
# bash tools/pmc_extra.sh <config>   -> gpurun_out/pmcx_c<config>.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
k=$1
cd /tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM" "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_VALU_TRANS_F64" "SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmcx_c${k}_$i -- python $R/tools/bench_configs.py --config $k --launches 2 > $R/gpurun_out/pmcx_c${k}_$i.log 2>&1
done
cd $R
python tools/rocpd_summary.py gpurun_out/pmcx_c${k}_* 2>&1 | grep "mc_\|PMC" | grep -v "^==" > gpurun_out/pmcx_c$k.txt
