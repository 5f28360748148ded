#!/bin/bash
# marginal cost of kernel families inside the overlapped step: skip their launches (results are garbage) and time the step.
# engine option ABLATE: entry points or "tag:<substring of a call-site tag>", '+' between entries inside one --cfg value.
run() { echo -n "$1 : "; python bench.py --no-cpu-baseline --no-extras --no-sweep --steps 4 --cfg "ablate=$1" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % d['ms_per_step'])"; }
run ""
run vinet_conv3d_wgrad
run vinet_bn_bwd_reduce
run vinet_bn_bwd_apply
run vinet_bn_bwd_reduce+vinet_bn_bwd_apply
run vinet_maxpool3d_bwd
run vinet_maxpool3d
run vinet_copy_affine
run tag:dgrad
run ""
