export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "tstream or stem_folded" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "split" 2>&1 | tail -3
B="python bench.py --dtype fp32s --batch 64 --steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-extras"
run() { echo "== $1"; env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for r in 1 2; do
run "VINET_OPT=conv_ts=0"
run "VINET_OPT=conv_ts=1"
done
