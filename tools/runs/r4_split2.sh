export TMPDIR=/tmp; mkdir -p gpurun_out
for o in 1 0; do
VINET_OPT=dma3=$o python bench.py --dtype fp32s --batch 64 --steps 3 --warmup 2 --no-sweep --no-extras --no-cpu-baseline --no-side-stream --profile-all 2> gpurun_out/r4_fp32s_serial_dma3_$o.sites | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32s b64 serial dma3=$o', round(d['value'],1), 'clips/s')"
head -8 gpurun_out/r4_fp32s_serial_dma3_$o.sites
done
