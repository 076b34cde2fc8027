# one line per configuration (tools/bench_configs.py) for profiles/<tag>_configs.jsonl
#   bash tools/run_configs_table.sh r04   (on the GPU box, from the repo root)
tag=${1:-r04}
out=gpurun_out/${tag}_configs.jsonl
: > $out
run() { env "$@" 2>/dev/null | tail -1 >> $out; }
for k in 1 3 4 5 6 7 8 9 10; do run python tools/bench_configs.py --config $k; done
run python tools/bench_configs.py --config 4 --replicas 2048
run python tools/bench_configs.py --config 5 --ladder 400,2000
run SMOLMC_FORCE_GENERAL=1 python tools/bench_configs.py --config 2
run SMOLMC_FORCE_UNIVERSAL=1 python tools/bench_configs.py --config 2 --mc 500
run SMOLMC_FORCE_UNIVERSAL=1 python tools/bench_configs.py --config 3 --mc 500
run SMOLMC_FORCE_UNIVERSAL=1 python tools/bench_configs.py --config 5 --mc 200
run SMOLMC_DENSE_EWALD=1 python tools/bench_configs.py --config 3 --mc 500
run SMOLMC_WL_PLAIN_ONLY=1 python tools/bench_configs.py --config 10
wc -l $out
