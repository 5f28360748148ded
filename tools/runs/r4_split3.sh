export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad" 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py -q -k "split_bf16 or fp32s or fp32" 2>&1 | grep -E "Error|rel err|diff|passed|failed|error" | head
python bench.py --dtype fp32s --batch 64 --steps 4 --warmup 2 --no-sweep --no-extras --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32s b64', round(d['value'],1), 'clips/s')"
python bench.py --dtype fp32s --batch 64 --steps 3 --warmup 2 --no-sweep --no-extras --no-cpu-baseline --no-side-stream --profile-all 2> gpurun_out/r4_fp32s_serial.sites | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32s b64 serial', round(d['value'],1), 'clips/s')"
head -8 gpurun_out/r4_fp32s_serial.sites
