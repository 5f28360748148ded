# in-place halo split (conv_ht SPLIT): kernel tests, then fp32s step A/B base lib vs new
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "halo_tile_split or exact or split_bf16" 2>&1 | tail -3
for r in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export VINET_LIB=$PWD/vinet_amd/libvinet_hip_base.so; else unset VINET_LIB; fi
  python bench.py --dtype fp32s --batch 64 --steps 4 --warmup 2 --no-sweep --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib fp32s', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
unset VINET_LIB
