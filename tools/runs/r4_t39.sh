export TMPDIR=/tmp; mkdir -p gpurun_out
for r in 1 2; do
for v in 0 1; do
  VINET_BRANCH_STREAMS_BWD=$v python bench.py --steps 4 --warmup 2 --sweep-steps 6 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['sweep']['local_batch_eager']; print('bwd=$v', round(d['value'],1), {k: round(v,1) for k,v in e.items()})"
done; done
