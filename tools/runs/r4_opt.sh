# alternating runs of option sets on one box: [BARGS="--batch 8"] bash tools/runs/r4_opt.sh rounds "A=1" "B=2,C=3" ...
export TMPDIR=/tmp; mkdir -p gpurun_out
R=$1; shift
for r in $(seq 1 $R); do
for o in "$@"; do
  VINET_OPT="$o" python bench.py --steps 8 --warmup 2 --no-sweep --no-extras --no-cpu-baseline $BARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$BARGS $o', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
