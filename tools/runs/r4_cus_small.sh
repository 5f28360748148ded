export TMPDIR=/tmp; mkdir -p gpurun_out
run() { echo "== $2 $1"; env $1 python bench.py --batch $2 --steps 10 --warmup 3 --no-sweep --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for b in 32 8; do
for r in 1 2; do
run "VINET_WGRAD_CUS=208" $b
run "VINET_WGRAD_CUS=128" $b
run "VINET_WGRAD_CUS=256" $b
run "VINET_DEFER_DECODER_WGRAD=0" $b
done; done
