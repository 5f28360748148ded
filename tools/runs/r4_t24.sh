export TMPDIR=/tmp; mkdir -p gpurun_out
for r in 1 2; do
for v in 0 1; do
for b in 1 2; do
  VINET_BRANCH_STREAMS_SWAP=$v python bench.py --mode infer --batch $b --graph --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph swap=$v batch=$b', round(d['value'],1), round(d['ms_per_step'],3))"
done; done; done
