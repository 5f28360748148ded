export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "temporal_frame_streaming" 2>&1 | tail -25
timeout 600 python tools/conv_ab.py --tm --batch 64 --rounds 5 --iters 3 2>&1 | grep -v amdgpu
