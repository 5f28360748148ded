#!/bin/bash
# A/B of the shared skip-gradient storage + model parity
cd /root/repo
python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3
for v in 0 1 0 1; do
  VINET_SHARE_SKIP_GRAD=$v python bench.py --no-sweep --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('share=$v', d['value'], d['ms_per_step'])"
done
