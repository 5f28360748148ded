# split-K convs that finish their own tiles (arrival counters) + decoder skip copies beside the conv: tests, batch-1/2/4 A/B
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "splitk" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "branch_streams or graphed_inference or harness or golden or e2e or infer" 2>&1 | tail -3
for r in 1 2; do
for o in "splitk_fused=0" "splitk_fused=1"; do
for b in 1 2 4; do
  VINET_OPT="$o" python bench.py --mode infer --batch $b --graph --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph $o batch=$b', round(d['value'],1), round(d['ms_per_step'],3))"
done; done; done
