export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -k "rowstream or exact" 2>&1 | tail -4
KERNELS="wgrad_rs" bash tools/runs/r4_ab.sh 1
