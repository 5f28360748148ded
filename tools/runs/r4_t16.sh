export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "halo_tile_split" 2>&1 | tail -3
B="python bench.py --dtype fp32s --batch 64 --steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-extras"
for r in 1 2; do $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"; done
