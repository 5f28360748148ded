# A/B of two builds of the library on one box: base = vinet_amd/libvinet_hip_base.so (VINET_LIB), new = the in-tree build.
# usage: bash tools/runs/r4_ab.sh [rounds]   -> serial (per-kernel site tables in gpurun_out/) and overlapped step times
export TMPDIR=/tmp; mkdir -p gpurun_out
for r in $(seq 1 ${1:-2}); do
for lib in base new; do
  if [ $lib = base ]; then export VINET_LIB=$PWD/vinet_amd/libvinet_hip_base.so; else unset VINET_LIB; fi
  python bench.py --steps 6 --warmup 2 --no-sweep --no-extras --no-cpu-baseline --no-side-stream --profile-all 2> gpurun_out/r4_ab_${lib}_serial.sites | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib serial', round(d['value'],1), round(d['ms_per_step'],2))"
  python bench.py --steps 8 --warmup 2 --no-sweep --no-extras --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib overlapped', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
unset VINET_LIB
for k in ${KERNELS:-wgrad_rs wgrad_tf wgrad_hs conv_hs wgrad_dma}; do echo "== $k"; grep " x[0-9]* *conv_$k\| x[0-9]* *$k" gpurun_out/r4_ab_base_serial.sites | head -8; echo "-- new"; grep " x[0-9]* *conv_$k\| x[0-9]* *$k" gpurun_out/r4_ab_new_serial.sites | head -8; done
