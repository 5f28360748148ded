export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad or split_bf16 or exact" 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py -q -k "split_bf16 or fp32s or weight_shared or mixed_joint or blocks_bf16 or train_step" 2>&1 | grep -E "Error|rel err|diff|passed|failed|error" | head
for o in 1 0; do VINET_SPLIT_WGRAD_BF16=$o python bench.py --dtype fp32s --batch 64 --steps 4 --warmup 2 --no-sweep --no-extras --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32s b64 split_wgrad_bf16=$o', round(d['value'],1), 'clips/s', d['config']['peak_hbm_gb'])"; done
python bench.py --dtype fp32s --batch 64 --steps 3 --warmup 2 --no-sweep --no-extras --no-cpu-baseline --no-side-stream --profile-all 2> gpurun_out/r4_fp32s_serial.sites > /dev/null
head -14 gpurun_out/r4_fp32s_serial.sites
