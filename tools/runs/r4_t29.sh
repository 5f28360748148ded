export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
for b in 1 2 4; do
  python bench.py --mode infer --batch $b --graph --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph batch=$b', round(d['value'],1), round(d['ms_per_step'],3))"
done
for r in 1 2; do
for o in "conv_ts_segs=0,conv_hs_segs=0" "conv_ts_segs=1,conv_hs_segs=1"; do
for b in 2 4 8; do
  VINET_OPT="$o" python bench.py --batch $b --steps 20 --warmup 3 --no-sweep --no-extras --no-cpu-baseline 2>gpurun_out/t29_err.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train $o batch=$b', round(d['value'],1), round(d['ms_per_step'],3))" || tail -5 gpurun_out/t29_err.log
done; done; done
