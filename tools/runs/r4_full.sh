export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
bash tools/prof_run.sh r4_v1 > gpurun_out/prof_r4_v1.log 2>&1
tail -3 gpurun_out/prof_r4_v1.log
cat gpurun_out/prof_r4_v1/bench.json | tail -c 3000
