# A/B of an environment switch on config 5 (from the repo root, on the GPU box): SWITCH=NAME runs the hot ladder
# (twice), BASELINE's cold ladder and 4096 walkers with the switch unset and set.  -> gpurun_out/ab_env5.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
out=gpurun_out/ab_env5.txt; : > $out
run() { python tools/bench_configs.py --config 5 $1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],3), round(d['acceptance'],4), d['kernel'])"; }
for sw in "" 1; do
  if [ -n "$sw" ]; then export $SWITCH=1; else unset $SWITCH; fi
  echo "== $SWITCH=$sw" >> $out
  for a in "" "" "--ladder 400,2000" "--replicas 4096"; do echo "config5 $a: $(run "$a")" >> $out; done
done
unset $SWITCH
cat $out
