export TMPDIR=/tmp; mkdir -p gpurun_out
for r in 1 2; do
for v in 0 200000; do
  VINET_BRANCH_STREAMS_TRAIN_VOX=$v python bench.py --steps 8 --warmup 2 --no-sweep --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b192 train_vox=$v', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
for v in 0 200000; do
for b in 4 16; do
  echo -n "train_vox=$v: "; VINET_BRANCH_STREAMS_TRAIN_VOX=$v python tools/sweep_small.py $b 10 2>/dev/null | tail -1
done; done
