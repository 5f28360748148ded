export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x --tb=short 2>&1 | grep -v "^$" | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), {k: round(v,1) for k,v in d['sweep']['local_batch_eager'].items()}, {k: round(v,1) for k,v in d['sweep']['local_batch'].items()})"
