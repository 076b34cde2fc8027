set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/r03_all_tests4.log 2>&1
grep -n "passed\|failed" gpurun_out/r03_all_tests4.log | tail -3
grep -n "^FAILED\|^ERROR" gpurun_out/r03_all_tests4.log | head
(time python bench.py --steps 20 --warmup 5) > gpurun_out/r03_bench_b.json 2> gpurun_out/r03_bench_b.err
tail -c 1500 gpurun_out/r03_bench_b.json; tail -5 gpurun_out/r03_bench_b.err
