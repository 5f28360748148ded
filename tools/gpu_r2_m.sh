#!/bin/bash
cd /root/repo
for v in 800 400 250 150 800 250; do
  VINET_MATERIALIZE_NT=$v python bench.py --no-sweep --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mat_nt=$v', d['value'], d['ms_per_step'], d['config'].get('peak_mem_gb'))"
done
