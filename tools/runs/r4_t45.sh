export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "training_forward_branch_streams" 2>&1 | tail -5
