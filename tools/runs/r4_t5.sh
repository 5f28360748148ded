export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "tstream or stem or exact_arithmetic or folded" 2>&1 | tail -8
bash tools/runs/r4_ab2.sh 3
