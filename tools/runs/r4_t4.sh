export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "train_step_split" 2>&1 | tail -3
B="python bench.py --dtype fp32s --batch 64 --steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-extras"
run() { echo "== $1"; env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run "X=1"
run "VINET_DEFER_DECODER_WGRAD_F32S=0"
run "VINET_DEFER_DECODER_WGRAD_F32S=0 VINET_SPLIT_ON_MAIN=1"
run "VINET_DEFER_DECODER_WGRAD_F32S=0 VINET_SPLIT_ON_MAIN=1 VINET_WGRAD_CUS=240"
run "VINET_DEFER_DECODER_WGRAD_F32S=0 VINET_SPLIT_ON_MAIN=1 VINET_WGRAD_CUS=176"
run "VINET_DEFER_DECODER_WGRAD_F32S=0 VINET_WGRAD_CUS=240"
run "VINET_DEFER_DECODER_WGRAD_F32S=1 VINET_SPLIT_ON_MAIN=1"
