export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -k "wgrad or stem or exact" 2>&1 | tail -4
python -m pytest tests/test_gpu_model.py -q -k "blocks_split or train_step_split" 2>&1 | grep -E "Error|rel err|diff|passed|failed" | head
bash tools/runs/r4_ab.sh 2
