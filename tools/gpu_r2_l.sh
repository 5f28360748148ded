#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "maxpool" 2>&1 | tail -3
for v in 3 1; do
  VINET_OPT="pool_twalk=$v" python bench.py --no-side-stream --profile-all --no-sweep --no-cpu-baseline --steps 2 --warmup 2 2>&1 >/dev/null | grep "maxpool_bwd_kernel |" | head -8
  echo "---- twalk=$v"
done
for v in 3 1 3 1; do
  VINET_OPT="pool_twalk=$v" python bench.py --no-sweep --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('twalk=$v', d['value'], d['ms_per_step'])"
done
