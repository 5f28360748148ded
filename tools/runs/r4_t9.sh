export TMPDIR=/tmp; mkdir -p gpurun_out
run() { echo "== $1"; env $1 python bench.py --steps 8 --warmup 2 --no-sweep --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for r in 1 2 3; do
run "VINET_TAIL_WGRAD_ON_MAIN=0"
run "VINET_TAIL_WGRAD_ON_MAIN=1"
done
VINET_TAIL_WGRAD_ON_MAIN=1 timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "weight_gradient_stream or graphed or train_step_bf16 or stem_bn" 2>&1 | tail -4
