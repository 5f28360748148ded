# split-K minimum chunks per split with branch streams on; n64 tile at batch 1; training small-batch A/B of sk_tile (repeat)
export TMPDIR=/tmp; mkdir -p gpurun_out
for o in "splitk=8" "splitk=12" "splitk=16" "splitk=20" "splitk=16,n64_tile=1" "splitk=16,n128_tile=1"; do
for b in 1 2 4; do
  VINET_OPT="$o" python bench.py --mode infer --batch $b --graph --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph $o batch=$b', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
for r in 1 2 3; do
for o in "sk_tile=0" "sk_tile=3"; do
for b in 1 2; do
  VINET_OPT="$o" python bench.py --batch $b --steps 20 --warmup 3 --no-sweep --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train $o batch=$b', round(d['value'],1), round(d['ms_per_step'],3))"
done; done; done
