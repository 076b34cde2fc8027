# Small Ewald lattices (fewer than nine groups of 64 changeable sites) with and without the masked small-lattice sweep
# (field_sweep_gx_small, round 6): the tree's library against a build with -DSMOLMC_NO_SMALL_SWEEP
# (smol_amd/exp/libsmolmc_tnosmall.so).  From the repo root on the GPU box.  -> gpurun_out/ab_small_cells.jsonl
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
out=gpurun_out/ab_small_cells.jsonl; : > $out
for f in smol_amd/exp/libsmolmc_tnosmall.so smol_amd/libsmolmc_hip.so; do
  for a in "3 --dim 6" "3 --dim 8" "9 --dim 8" "10 --dim 8" "13 --dim 8" "11 --dim 4" "11 --dim 6" "3 --dim 12" "11 --dim 8"; do
    SMOLMC_LIB=$PWD/$f python tools/bench_configs.py --config $a --launches 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(dict(library=\"$(basename $f)\", args=\"$a\", kernel_ms=round(d[\"kernel_ms\"],4), mc_steps_per_s=d[\"mc_steps_per_s\"], acceptance=round(d[\"acceptance\"],4), kernel=d[\"kernel\"][:64])))" >> $out
  done
done
cat $out
