export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x --tb=short 2>&1 | grep -v "^$" | tail -60 > gpurun_out/t48.log
tail -45 gpurun_out/t48.log | cut -c1-250
