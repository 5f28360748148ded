# Inception branches over three streams in small-batch inference: test, then batch-1 / 2 / 4 fps with and without (hipGraph and eager)
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "branch_streams or graphed_inference or harness" 2>&1 | tail -3
for r in 1 2; do
for v in 0 65536; do
for b in 1 2 4; do
  VINET_BRANCH_STREAMS_VOX=$v python bench.py --mode infer --batch $b --graph --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph vox=$v batch=$b', round(d['value'],1), round(d['ms_per_step'],3))"
done; done; done
for v in 0 65536; do
  VINET_BRANCH_STREAMS_VOX=$v python bench.py --mode infer --batch 1 --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager vox=$v batch=1', round(d['value'],1), round(d['ms_per_step'],3))"
done
