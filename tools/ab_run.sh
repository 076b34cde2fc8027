#!/bin/bash
# run bench.py (kernel time only) for every variant library under smol_amd/exp, twice, interleaved
for rep in 1 2; do
for f in smol_amd/exp/libsmolmc_*.so; do
  t=$(SMOLMC_LIB=$PWD/$f python bench.py --no-cpu-baseline --steps 6 --warmup 2 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['acceptance_ratio'])")
  echo "$(basename $f) $t"
done; done
