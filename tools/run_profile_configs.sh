# rocprofv3 passes for the non-headline configurations (tools/bench_configs.py): kernel-trace
# stats, then separate PMC passes.  Writes gpurun_out/<tag>_configs_pmc.txt and merges the per-config
# constants (VALU / LDS issue fractions, HBM bytes per launch, stamped with the source digest) into
# gpurun_out/pmc_constants_<tag>.json (started by tools/run_profile.sh).
#   bash tools/run_profile_configs.sh r04        (on the GPU box, from the repo root)
set -x
tag=${1:-r04}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFGS=${CFGS:-"3 r3 9 4 5 10 11 12 13 d3 u2 l13"}
# config -> bench_configs arguments | kernel name fragment | replicas | steps per launch
args_of() { case $1 in
  3) echo "--config 3";; r3) echo "--config 3 --temperature 3000";; 9) echo "--config 9";; 4) echo "--config 4 --mc 20000";; 5) echo "--config 5";;
  10) echo "--config 10";; 11) echo "--config 11";; 12) echo "--config 12";; 13) echo "--config 13";; d3) echo "--config 3 --mc 500";; u2) echo "--config 2 --mc 2000";; l13) echo "--config 13";; esac; }
kern_of() { case $1 in 3|r3|9|12|13) echo "mc_lean_kernel";; 11) echo "mc_lean_multi_kernel";; 4|10) echo "mc_wl_kernel";; 5) echo "mc_table_kernel";; d3) echo "mc_kernel";; u2) echo "mc_univ_kernel";; l13) echo "mc_lean_kernel";; esac; }
reps_of() { case $1 in 4|10|11) echo 1024;; u2|12) echo 4096;; *) echo 2048;; esac; }
mc_of() { case $1 in 3|r3|9|10|11|13|u2|l13) echo 2000;; 12) echo 10000;; 4) echo 20000;; 5) echo 3456;; d3) echo 500;; esac; }
cd /tmp
for k in $CFGS; do
  pre=""; [ $k = d3 ] && pre="SMOLMC_DENSE_EWALD=1"; [ $k = u2 ] && pre="SMOLMC_FORCE_UNIVERSAL=1"; [ $k = l13 ] && pre="SMOLMC_LAZY_FEATURES_ONLY=1"
  env $pre rocprofv3 --kernel-trace --stats -d $R/gpurun_out/cprof_${tag}_c$k -- python $R/tools/bench_configs.py $(args_of $k) > $R/gpurun_out/cprof_${tag}_c$k.log 2>&1
  i=0
  for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    env $pre rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/cpmc_${tag}_c${k}_$i -- python $R/tools/bench_configs.py $(args_of $k) --launches 2 > $R/gpurun_out/cpmc_${tag}_c${k}_$i.log 2>&1
  done
done
cd $R
cp gpurun_out/pmc_constants_${tag}.json gpurun_out/pmc_constants_${tag}_all.json 2>/dev/null || echo "{}" > gpurun_out/pmc_constants_${tag}_all.json
for k in $CFGS; do
  echo "######## config $k: $(args_of $k)"; tail -1 gpurun_out/cprof_${tag}_c$k.log
  python tools/rocpd_summary.py gpurun_out/cprof_${tag}_c$k gpurun_out/cpmc_${tag}_c${k}_*
  key=config$k; [ $k = r3 ] && key=config3_reject_path; [ $k = d3 ] && key=config3_dense_ewald; [ $k = u2 ] && key=config2_universal; [ $k = l13 ] && key=config13_lazy
  python tools/pmc_to_json.py --kernel "$(kern_of $k)" --replicas $(reps_of $k) --mc $(mc_of $k) --key $key \
     --source profiles/${tag}_configs_pmc.txt --merge gpurun_out/pmc_constants_${tag}_all.json gpurun_out/cpmc_${tag}_c${k}_* > gpurun_out/pmc_tmp.json \
     && mv gpurun_out/pmc_tmp.json gpurun_out/pmc_constants_${tag}_all.json
done > gpurun_out/${tag}_configs_pmc.txt 2>&1
rm -rf gpurun_out/cprof_${tag}_c*/ gpurun_out/cpmc_${tag}_c*/
