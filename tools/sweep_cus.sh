for cus in 96 128 192 224; do
  echo "WGRAD_CUS=$cus"; VINET_WGRAD_CUS=$cus python bench.py --no-cpu-baseline --no-extras --no-sweep --steps 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
echo "no defer"; VINET_DEFER_DECODER_WGRAD=0 python bench.py --no-cpu-baseline --no-extras --no-sweep --steps 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "no side stream"; python bench.py --no-cpu-baseline --no-extras --no-sweep --steps 6 --no-side-stream 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "default"; python bench.py --no-cpu-baseline --no-extras --no-sweep --steps 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
