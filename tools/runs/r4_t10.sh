export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "decoder or train_step or e2e_bf16 or weight_gradient_stream or graphed" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "upsample" 2>&1 | tail -3
run() { echo "== $1"; env $1 python bench.py --steps 8 --warmup 2 --no-sweep --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for r in 1 2 3; do
run "VINET_UPSAMPLE_BWD_RELU=0"
run "VINET_UPSAMPLE_BWD_RELU=1"
done
