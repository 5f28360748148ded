export TMPDIR=/tmp; mkdir -p gpurun_out
run() { echo "== $1"; env $1 python bench.py --steps 8 --warmup 2 --no-sweep --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), d['config']['peak_hbm_gb'])"; }
for r in 1 2; do
run "X=1"
run "VINET_OPT=ht_pre=1"
run "VINET_OPT=ht_pre=1 VINET_SIDE_MATERIALIZE=1"
done
