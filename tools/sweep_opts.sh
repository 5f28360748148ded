#!/bin/bash
# whole-step A/B of tuning switches: each line "ENV=VALUE ..." runs the headline bench (6 steps) and prints clips/s
run() { echo -n "$* : "; env "$@" python bench.py --no-cpu-baseline --no-extras --no-sweep --steps 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f clips/s  %.2f ms' % (d['value'], d['ms_per_step']))"; }
run X=0
while read -r line; do [ -n "$line" ] && run $line; done
run X=0
