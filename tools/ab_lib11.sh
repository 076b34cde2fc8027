# A/B of library variants on config 11 (LiNiO2 8^3 + Ewald under Wang-Landau; from the repo root, on the GPU box): the tree's
# library and smol_amd/exp/libsmolmc_t*.so, swaps at 1024 and 4096 walkers.  -> gpurun_out/ab_lib11.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
out=gpurun_out/ab_lib11.txt; : > $out
for f in smol_amd/libsmolmc_hip.so smol_amd/exp/libsmolmc_t*.so; do
  [ -e $f ] || continue
  for a in "--replicas 1024" "--replicas 1024" "--replicas 4096"; do
    SMOLMC_LIB=$PWD/$f python tools/bench_configs.py --config 11 $a --launches 3 2>/dev/null | tail -1 > /tmp/o.txt
    echo "$(basename $f) config11 $a: $(python -c "import json; d=json.loads(open('/tmp/o.txt').read()); print(round(d['kernel_ms'],3), '%.3e' % d['mc_steps_per_s'], round(d['acceptance'],4), d['kernel'])")" >> $out
  done
done
cat $out
