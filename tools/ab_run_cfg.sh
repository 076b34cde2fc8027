#!/bin/bash
# A/B of the variant libraries under smol_amd/exp on one tools/bench_configs.py configuration:
# tools/ab_run_cfg.sh <config> [extra args]
c=$1; shift
for rep in 1 2; do
for f in smol_amd/exp/libsmolmc_*.so; do
  t=$(SMOLMC_LIB=$PWD/$f python tools/bench_configs.py --config $c "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d.get('kernel_ms', d.get('kernel_ms_last')), d['acceptance'])")
  echo "$(basename $f) $t"
done; done
