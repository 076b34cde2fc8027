for f in smol_amd/exp/libsmolmc_t*.so; do
  echo "$(basename $f) $(SMOLMC_LIB=$PWD/$f python tools/bench_configs.py --config 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],3), round(d['acceptance'],4))")"
done
