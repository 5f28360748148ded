export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "tstream or stem or conv3d" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x 2>&1 | tail -2
for r in 1 2; do
for b in 1 2 4; do
  python bench.py --mode infer --batch $b --graph --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph batch=$b', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
