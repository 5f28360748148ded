#!/bin/bash
# whole-step A/B of engine / library options on ONE box: every line of stdin is a --cfg string
# ("wgrad_cus=192", "lib.ht_pre=1,materialize_nt=400", "" = defaults); the headline bench (6 steps) runs once per line,
# with the default configuration before and after (boxes and clocks drift: compare neighbours, not runs of different calls).
#   printf 'wgrad_cus=176\nwgrad_cus=224\n' | tools/sweep_cfg.sh [extra bench.py flags]
run() { echo -n "cfg='$1' : "; python bench.py --no-cpu-baseline --no-extras --no-sweep --steps 6 --cfg "$1" "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f clips/s  %.2f ms' % (d['value'], d['ms_per_step']))"; }
run "" "$@"
while read -r line; do run "$line" "$@"; done
run "" "$@"
