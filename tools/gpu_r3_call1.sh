set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/equil_sweep.py --config 3 --T 3000 6000 12000 --mu 0.5 0.1 0.02 --equil 600000 > gpurun_out/r03_sweep3.jsonl 2> gpurun_out/r03_sweep3.err
python tools/equil_sweep.py --config 5 --T 400:2000 1000:5000 2000:10000 --mu 0.5 0.1 0.02 --equil 400000 > gpurun_out/r03_sweep5.jsonl 2> gpurun_out/r03_sweep5.err
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/r03_gputests0.log 2>&1
tail -3 gpurun_out/r03_gputests0.log
cat gpurun_out/r03_sweep3.jsonl gpurun_out/r03_sweep5.jsonl | cut -c1-400
