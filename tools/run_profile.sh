# Round profile of the headline benchmark: bench line, rocprofv3 kernel stats, five PMC passes
# (separate runs: --pmc must not be combined with tracing domains other than --kernel-trace).
#   bash tools/run_profile.sh r02       (on the GPU box, from the repo root)
# Outputs under gpurun_out/ (scratch); copy the summaries you want judged into profiles/.
set -x
tag=${1:-r03}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/bench_${tag}.json 2> gpurun_out/bench_${tag}.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag} -- python $R/bench.py --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_${tag}.log 2>&1
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_${tag}_$n -- python $R/bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 > $R/gpurun_out/pmc_${tag}_$n.log 2>&1
done
# LDS pipe of the headline kernel (index-active / conflict cycles, FIFO stalls, shader clock)
i=0
for c in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/ldspmc_${tag}_$i -- python $R/bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 > $R/gpurun_out/ldspmc_${tag}_$i.log 2>&1
done
cd $R
python tools/rocpd_summary.py gpurun_out/ldspmc_${tag}_* 2>&1 | grep -v "not a database\|\.log" > gpurun_out/${tag}_headline_lds_pmc.txt
python tools/rocpd_summary.py gpurun_out/prof_${tag} gpurun_out/pmc_${tag}_* 2>&1 | grep -v "not a database\|\.log" > gpurun_out/${tag}_headline_pmc.txt
python tools/pmc_to_json.py --kernel "mc_lean_kernel<2, 2, 1, false, 0, false, false, true" --replicas 4096 --mc 125000 --source profiles/${tag}_headline_pmc.txt gpurun_out/pmc_${tag}_* > gpurun_out/pmc_constants_${tag}.json
tail -c 600 gpurun_out/bench_${tag}.json
# the raw rocpd databases are scratch (gpurun merges at most 64 MiB back): keep the summaries only
rm -rf gpurun_out/prof_${tag} gpurun_out/pmc_${tag}_*/ gpurun_out/ldspmc_${tag}_*/
