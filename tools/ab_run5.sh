#!/bin/bash
for f in smol_amd/exp/libsmolmc_*.so; do
  t=$(SMOLMC_LIB=$PWD/$f python tools/bench_configs.py --config 5 --launches 4 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms_last'], d['acceptance'])")
  echo "$(basename $f) $t"
done
