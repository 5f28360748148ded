export TMPDIR=/tmp; mkdir -p gpurun_out
run() { echo "== $1"; env $1 python bench.py --steps 8 --warmup 2 --no-sweep --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for r in 1 2; do
run "VINET_WGRAD_CUS=208"
run "VINET_WGRAD_CUS=224"
run "VINET_WGRAD_CUS=192"
run "VINET_WGRAD_CUS=176"
run "VINET_WGRAD_CUS=208 VINET_WGRAD_CUS_DEC=160"
done
