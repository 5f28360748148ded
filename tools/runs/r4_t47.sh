export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_model.py -q -m gpu -x 2>&1 | grep -v "^$" | tail -40 > gpurun_out/t47.log
tail -40 gpurun_out/t47.log
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "training_forward_branch_streams" 2>&1 | tail -1; done
