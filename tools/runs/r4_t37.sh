export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "training_forward_branch_streams or weight_gradient_stream or trajectory or graphed_train" 2>&1 | tail -3
for r in 1 2; do
for v in 0 1; do
for b in 4 8 16 32; do
  echo -n "bwd_fork=$v: "; VINET_BRANCH_STREAMS_BWD=$v python tools/sweep_small.py $b 10 2>/dev/null | tail -1
done; done; done
