# A/B of the TableFlip kernels on config 5 and the LiNiO2 model (from the repo root, on the GPU box): the library of
# the tree and the variants smol_amd/exp/libsmolmc_t*.so (tools/build_variant.sh).  Per library: the TableFlip parity
# tests through SMOLMC_LIB, config 5 on the hot ladder (twice), on BASELINE's cold ladder and at 4096 walkers, LiNiO2
# 8^3 / 12^3.  The tree's library also runs the TableFlip stress cases.  -> gpurun_out/ab_table5.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
out=gpurun_out/ab_table5.txt; : > $out
run() { SMOLMC_LIB=$1 python tools/bench_configs.py --config 5 $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],3), round(d['acceptance'],4))"; }
for f in smol_amd/libsmolmc_hip.so smol_amd/exp/libsmolmc_t*.so; do
  [ -e $f ] || continue
  echo "== $f" >> $out
  SMOLMC_LIB=$PWD/$f timeout 600 python -m pytest tests/test_gpu_table_flip.py tests/test_gpu_replay_v6.py -x -q -m gpu 2>&1 | tail -2 >> $out
  for a in "" "" "--ladder 400,2000" "--replicas 4096"; do echo "config5 $a: $(run $PWD/$f "$a")" >> $out; done
  if [ -z "$NO_MSON" ]; then
    for d in 8 12; do SMOLMC_LIB=$PWD/$f python tools/bench_mson.py --dim $d 2>/dev/null | grep table-flip | cut -c1-400 >> $out; done
  fi
done
timeout 600 python tests/stress_table_flip.py > gpurun_out/ab_table5_stress.txt 2>&1
echo "stress rc=$?" >> $out; tail -4 gpurun_out/ab_table5_stress.txt >> $out
cat $out
