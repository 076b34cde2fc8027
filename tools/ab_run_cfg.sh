#!/bin/bash
# kernel time of tools/bench_configs.py configurations for every variant library under smol_amd/exp
# usage: bash tools/ab_run_cfg.sh "1 3 4 6" ["--replicas 16384"]
for rep in 1 2; do
for k in $1; do
for f in smol_amd/exp/libsmolmc_*.so; do
  t=$(SMOLMC_LIB=$PWD/$f python tools/bench_configs.py --config $k $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],4), d['acceptance'])")
  echo "config $k $2 $(basename $f) $t"
done; done; done
