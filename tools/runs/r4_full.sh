export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/prof_run.sh r4_v7 > gpurun_out/prof_r4_v7.log 2>&1
tail -2 gpurun_out/prof_r4_v7.log
