# A/B of two builds on one box, overlapped step only, alternating: base = vinet_amd/libvinet_hip_base.so, new = in-tree
export TMPDIR=/tmp; mkdir -p gpurun_out
for r in $(seq 1 ${1:-3}); do
for lib in base new; do
  if [ $lib = base ]; then export VINET_LIB=$PWD/vinet_amd/libvinet_hip_base.so; else unset VINET_LIB; fi
  python bench.py --steps 8 --warmup 2 --no-sweep --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib overlapped', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
unset VINET_LIB
