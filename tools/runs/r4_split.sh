export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -k "split_bf16" 2>&1 | tail -5
python -m pytest tests/test_gpu_model.py -q -k "split_bf16 or fp32s" 2>&1 | grep -E "Error|rel err|diff|passed|failed|error" | head
python bench.py --dtype fp32s --batch 64 --steps 4 --warmup 2 --no-sweep --no-extras --no-cpu-baseline --profile-all 2> gpurun_out/r4_fp32s_b64_dma3.sites | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32s b64', round(d['value'],1), 'clips/s')"
head -24 gpurun_out/r4_fp32s_b64_dma3.sites
