export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -k "wgrad or stem or exact" 2>&1 | tail -3
KERNELS="wgrad_rs wgrad_tf wgrad_hs wgrad_ts wgrad_dma" bash tools/runs/r4_ab.sh 2
