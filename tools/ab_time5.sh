# timing-only A/B of mc_table_kernel variants on config 5 (from the repo root, on the GPU box): the library of the tree and
# smol_amd/exp/libsmolmc_t*.so (tools/build_variant.sh), hot ladder (twice) and BASELINE's cold ladder; phase lines of a variant
# built with -DSMOLMC_EXP_PHASES are kept.  ENVS="A=1 B=1" exports switches for every run.  -> gpurun_out/ab_time5.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
out=gpurun_out/ab_time5.txt; : > $out
for e in $ENVS; do export $e; done
for f in smol_amd/libsmolmc_hip.so smol_amd/exp/libsmolmc_t*.so; do
  [ -e $f ] || continue
  for a in "" "" "--ladder 400,2000"; do
    SMOLMC_LIB=$PWD/$f python tools/bench_configs.py --config 5 $a --launches ${LAUNCHES:-3} > /tmp/o.txt 2>/dev/null
    echo "$(basename $f) config5 $a: $(tail -1 /tmp/o.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],3), round(d['acceptance'],4))")" >> $out
    grep -a "phases\|batch:\|proposal:" /tmp/o.txt | tail -6 >> $out
  done
done
cat $out
