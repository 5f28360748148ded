export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "tstream or stem or folded" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "train_step or e2e or graphed or weight_gradient_stream" 2>&1 | tail -6
run() { echo "== $1"; env $1 python bench.py --steps 8 --warmup 2 --no-sweep --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for r in 1 2 3; do
run "VINET_DGRAD_BN_STATS=0"
run "VINET_DGRAD_BN_STATS=1"
done
