export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "training_forward_branch_streams" 2>&1 | tail -3
for r in 1 2; do
for v in 0 16384 32768 200000; do
for b in 8 32; do
  echo -n "train_vox=$v: "; VINET_BRANCH_STREAMS_TRAIN_VOX=$v python tools/sweep_small.py $b 10 2>/dev/null | tail -1
done; done; done
